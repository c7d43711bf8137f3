import sys, traceback
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from tests.test_gpu_fuzz import test_random_configuration as f
gpu = torch.device("cuda", 0)
bad = []
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 212)
for seed in range(lo, hi):
    try:
        f(gpu, seed)
    except Exception as e:
        tb = traceback.extract_tb(sys.exc_info()[2])
        where = next((f"{os.path.basename(fr.filename)}:{fr.lineno} `{(fr.line or '').strip()[:110]}`" for fr in reversed(tb)
                      if fr.filename.endswith("test_gpu_fuzz.py")), "")
        bad.append((seed, repr(e)[:300]))
        print("FAIL", seed, repr(e)[:300], "AT", where, flush=True)  # which assertion: the column of fuzz_diag.py to read
    if seed % 25 == 0: print("done", seed, flush=True)
print("failures:", len(bad), bad[:5])
