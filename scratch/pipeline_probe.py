"""shape cycling vs loader: (a) 8 distinct mixed-length batches RESIDENT on the device, cycled; (b) the same through DevicePrefetcher from host tensors prepared up front"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mm-dti_amd"))
import torch, bench
from mmdti_hip.trainer import FineTuner
from mmdti_hip.collate import device_payload, to_device
from mmdti_hip.data import DevicePrefetcher
model, _ = bench.build_model()
model = model.cuda().train()
tuner = FineTuner(model, "classification", total_steps=10000)
host = []
for i in range(8):
    _, b, y = bench.synth(256, 128, 256, seed=100 + i, ragged=True)
    host.append((device_payload(b, 961, 0), y))
dev = [(to_device(b, "cuda"), y.cuda()) for b, y in host]
def loop(get, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = 0.0
    for i in range(n):
        b, y = get(i)
        h0 = time.perf_counter()
        tuner.step(b, y, epoch=0)
        th += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3
for rep in range(2):
    print("resident, one batch      ms/step %.1f  host in step %.1f" % loop(lambda i: dev[0], 16))
    print("resident, 8 shapes cycled ms/step %.1f  host in step %.1f" % loop(lambda i: dev[i % 8], 16))
    print("peak alloc GB", torch.cuda.max_memory_allocated() / 1e9, "reserved", torch.cuda.memory_reserved() / 1e9)
def pf_loop(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b, y in DevicePrefetcher((host[i % 8] for i in range(n)), "cuda", narrow=False):
        tuner.step(b, y, epoch=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("prefetcher from prepared host batches ms/step %.1f" % pf_loop(16))
