"""Where does the host time of the fresh-batch-every-step loop go?  cProfile over 16 steps of bench.py's pipeline workload."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench


def main():
    from types import SimpleNamespace
    from mmdti_hip.trainer import FineTuner
    model, _ = bench.build_model()
    model = model.cuda().train()
    tuner = FineTuner(model, "classification", total_steps=10000)
    args = SimpleNamespace(batch=256, atoms=128, tokens=256, steps=16)
    dev = torch.device("cuda", 0)
    def barrier(): torch.cuda.synchronize()
    r = bench.pipeline_workload(tuner, model, args, dev, 1, 0, barrier)
    print({k: v for k, v in r.items() if k not in ("workload", "note")})
    pr = cProfile.Profile()
    pr.enable()
    r = bench.pipeline_workload(tuner, model, args, dev, 1, 0, barrier)
    pr.disable()
    print({k: v for k, v in r.items() if k not in ("workload", "note")})
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
