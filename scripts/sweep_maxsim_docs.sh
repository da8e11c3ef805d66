# pass-1 form x documents per block, on UCC-en (ucc_colbert) and zh + en (full_hybrid_rerank): gpurun_out/r4/ms_sweep.txt
O=gpurun_out/r4/ms_sweep
mkdir -p $O
for m in ${FORMS:-ring pipe}; do for d in ${DOCS:-8 16 32 64 128}; do
  AMDR_MAXSIM_PASS1=$m AMDR_MAXSIM_DOCS=$d timeout -k 10 120 python bench.py --only ucc_colbert --steps 10 > $O/en_${m}_$d.json 2>/dev/null
  AMDR_MAXSIM_PASS1=$m AMDR_MAXSIM_DOCS=$d timeout -k 10 120 python bench.py --only full_hybrid_rerank --steps 10 > $O/zh_${m}_$d.json 2>/dev/null
  python - <<PY >> gpurun_out/r4/ms_sweep.txt
import json
a=json.loads(open("$O/en_${m}_$d.json").read().strip().splitlines()[-1])["ucc_colbert"]
b=json.loads(open("$O/zh_${m}_$d.json").read().strip().splitlines()[-1])["full_hybrid_rerank"]
print("$m docs=$d  en step ms", round(a["timing"]["median"],4), " zh+en step ms", round(b["timing"]["median"],4), " zh maxsim ms", round(b["per_lang"]["zh"]["maxsim_ms"],4), " en maxsim ms", round(b["per_lang"]["en"]["maxsim_ms"],4))
PY
done; done
cat gpurun_out/r4/ms_sweep.txt
