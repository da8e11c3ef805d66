"""Retrieval path: same module / class names as legalrag.retrieval."""
