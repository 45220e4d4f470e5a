"""GPU micro-benchmarks of the individual kernels (HIP-event timing of back-to-back launches)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nf-isam_amd")); sys.path.insert(0, ROOT)
import nfisam_hip as nh
import bench as BM

dev = torch.device("cuda:0")
K, H, B = 9, 8, 5.0


def timeit(fn, reps=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def problem(n, D, L, seed=0):
    rng = np.random.RandomState(seed)
    x = torch.from_numpy(rng.randn(n, D).astype(np.float32)).to(dev)
    blob = torch.from_numpy(BM.init_blob_np(D, K, H, L, seed)).to(dev)
    return x, nh.pack(blob, D, K, H, L)


def main():
    lib = nh.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("%-44s %10s" % ("case", "us/launch"))
    for (n, D, L) in [(4096, 6, 1), (4096, 6, 2), (4096, 6, 4), (2000, 15, 1), (2000, 11, 1), (64, 6, 4), (64, 6, 1),
                      (65536, 6, 1), (65536, 15, 1), (262144, 6, 4)]:
        x, kp = problem(n, D, L)
        z = torch.empty_like(x); ld = torch.empty(n, device=dev)
        g = torch.zeros_like(kp); loss = torch.zeros(1, device=dev)
        t_f = timeit(lambda: lib.nfisam_nsf_forward(C.c_void_p(x.data_ptr()), C.c_void_p(kp.data_ptr()), n, D, K, H,
                                                    C.c_float(B), L, C.c_size_t(0), C.c_void_p(z.data_ptr()),
                                                    C.c_void_p(ld.data_ptr()), None, st))
        t_b = timeit(lambda: lib.nfisam_nsf_backward(C.c_void_p(x.data_ptr()), C.c_void_p(kp.data_ptr()), n, D, K, H,
                                                     C.c_float(B), L, C.c_size_t(0), None, None, 1,
                                                     C.c_void_p(g.data_ptr()), None, C.c_void_p(loss.data_ptr()), st))
        t_i = timeit(lambda: lib.nfisam_nsf_inverse(C.c_void_p(z.data_ptr()), None, C.c_void_p(kp.data_ptr()), n, D, 0, K,
                                                    H, C.c_float(B), L, C.c_size_t(0), None, None, None,
                                                    C.c_void_p(x.data_ptr()), None, st))
        fl = BM.flops_per_sample_iter(D, K, H, L) * n
        print("n=%-7d D=%-3d L=%d  forward %9.2f  train %9.2f (%.2f TFLOP/s)  inverse %9.2f" %
              (n, D, L, t_f, t_b, fl / t_b / 1e6, t_i))
    # batched training step: nc cliques of the scaling shape
    for (nc, n, D, L) in [(1, 2000, 15, 1), (8, 2000, 15, 1), (64, 2000, 15, 1), (256, 2000, 15, 1), (64, 4096, 6, 4)]:
        xs, kps = [], []
        for c in range(nc):
            x, kp = problem(n, D, L, seed=c)
            xs.append(x); kps.append(kp)
        tb = nh.TrainBatch(xs, kps, K, H, B, L, lr=0.01, max_iters=100000, early_stop=False)
        t = timeit(tb.step, reps=50, warm=5)
        fl = BM.flops_per_sample_iter(D, K, H, L) * n * nc
        print("batched nc=%-4d n=%-5d D=%-3d L=%d  step %9.2f us  %.3e samples/s  %.2f TFLOP/s" %
              (nc, n, D, L, t, nc * n / t * 1e6, fl / t / 1e6))


if __name__ == "__main__":
    main()
