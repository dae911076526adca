"""bench.py's pipeline workload under the four combinations of (DataLoader pin_memory, DevicePrefetcher threaded)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "mm-dti_amd"))
import torch, bench
from types import SimpleNamespace
from mmdti_hip.trainer import FineTuner
def main():
    model, _ = bench.build_model()
    model = model.cuda().train()
    tuner = FineTuner(model, "classification", total_steps=10000)
    args = SimpleNamespace(batch=256, atoms=128, tokens=256, steps=24)
    r = bench.pipeline_workload(tuner, model, args, torch.device("cuda", 0), 1, 0, torch.cuda.synchronize)
    print(os.environ.get("MMDTI_BENCH_PIN"), os.environ.get("MMDTI_BENCH_THREADED"), {k: v for k, v in r.items() if k not in ("workload", "note")}, flush=True)
if __name__ == "__main__":
    main()
''' % (ROOT, ROOT)
open("/tmp/pm.py", "w").write(code)
for pin in ("0", "1"):
    for thr in ("0", "1"):
        subprocess.run([sys.executable, "/tmp/pm.py"], env=dict(os.environ, MMDTI_BENCH_PIN=pin, MMDTI_BENCH_THREADED=thr))
