"""debug: multi-tile passes of the round-4 tail"""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from legal_rag_amd import _native
rng = np.random.default_rng(3)
n, d, nq = 200_000, 256, 256
X = rng.standard_normal((n, d)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
Q = rng.standard_normal((nq, d)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
os.environ["AMDR_DENSE_HI"] = "1"; os.environ["AMDR_DENSE_TWO_LEVEL"] = "1"
def run(tag, q, **env):
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    idx = _native.DenseIndex(X)
    s, i = idx.search(q, 10)
    print(tag, len(q), idx.hi_counters(), flush=True)
    idx.close()
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    return i
cap = str(64 * 1500)
ref = run("exact", Q, AMDR_DENSE_HI="0")
for m in (64, 128, 192, 256):
    a = run(f"one   cap1500 m={m}", Q[:m], AMDR_DENSE_HI_CAP=cap)
    b = run(f"split cap1500 m={m}", Q[:m], AMDR_DENSE_HI_CAP=cap, AMDR_DENSE_HI_SCANS="split")
    c = run(f"one   nocap   m={m}", Q[:m])
    print("   equal to exact:", np.array_equal(a, ref[:m]), np.array_equal(b, ref[:m]), np.array_equal(c, ref[:m]))
run("second half alone (T=1)", Q[64:128], AMDR_DENSE_HI_CAP=cap)
run("old tail m=128", Q[:128], AMDR_DENSE_HI_TAIL="0")
