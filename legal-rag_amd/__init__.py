"""MI355X-native hybrid retrieval engine — drop-in for legalrag.retrieval.

Host-side mirror of the reference's retriever API (HybridRetriever.search /
RetrievalHit, same index artifact layout) over libamdretrieval.so
(hand-written HIP kernels for gfx950, C ABI in include/amdretrieval.h).
Import as `legal_rag_amd` (the on-disk directory name `legal-rag_amd` is not a
valid Python identifier; `legal_rag_amd/__init__.py` at the repo root aliases it).
"""
__version__ = "0.1.0"
