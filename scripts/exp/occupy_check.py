"""Diagnostic (GPU): is the library's device occupier (nfisam_diag_occupy_device, libnfisam_diag.so) on the machine, and what does the probe say then?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import torch
import nfisam_hip as nh
os.environ["NFISAM_PROBE_DEBUG"] = "1"
dev = torch.device("cuda", 0)
lib = nh.lib()
diag = ctypes.CDLL(os.path.join(os.path.dirname(nh.LIB_PATH), "libnfisam_diag.so"))
cus = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs", cus, "wall clock rate attr:", end=" ")
try:
    print(torch.cuda.get_device_properties(0))
except Exception as e:
    print(e)
side = torch.cuda.Stream()
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
for blocks, lds, sec in ((cus, 100 * 1024, 0.5), (cus, 100 * 1024, 1.5), (16 * cus, 20 * 1024, 0.5)):
    t0 = time.perf_counter()
    rc = diag.nfisam_diag_occupy_device(blocks, ctypes.c_size_t(lds), ctypes.c_float(sec), ctypes.c_void_p(side.cuda_stream))
    t1 = time.perf_counter()
    y = x + 1                       # a small kernel on the default stream while the occupier runs
    torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    side.synchronize()
    t3 = time.perf_counter()
    print("occupier blocks %d lds %d sec %.1f: rc %d, launch %.1f ms, small kernel done after %.1f ms, occupier done after %.1f ms" %
          (blocks, lds, sec, rc, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t3 - t0)), flush=True)
# the probe while the occupier is on: a C3-shaped plan
K, H, B = 9, 8, 5.0
gen = torch.Generator().manual_seed(5)
shapes = [(2000, D) for D in (6, 8, 8, 10, 10, 12, 12, 12)]
xs = [(1.2 * torch.randn(n, D, generator=gen)).clamp_(-4, 4).to(dev) for n, D in shapes]
kp0 = [nh.pack((0.2 * torch.randn(nh.param_count(D, K, H), generator=gen)).to(dev), D, K, H, 1) for n, D in shapes]
for occupy in (False, True):
    time.sleep(0.7)
    if occupy:
        diag.nfisam_diag_occupy_device(cus, ctypes.c_size_t(100 * 1024), ctypes.c_float(1.5), ctypes.c_void_p(side.cuda_stream))
        time.sleep(0.05)
    tb = nh.TrainBatch(xs, [k.clone() for k in kp0], K, H, B, 1, lr=0.01, max_iters=100, average_window=50, loss_delta_tol=0.0, early_stop=True)
    t0 = time.perf_counter()
    tb.prepare(True)
    print("occupied" if occupy else "quiet", "prepare %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
    try:
        it = tb.run(); torch.cuda.synchronize()
        print("  ran", it, "span", tb.xcd_span(), flush=True)
    except Exception as e:
        print("  run failed:", repr(e)[:200], flush=True)
    tb.close()
    torch.cuda.synchronize()
