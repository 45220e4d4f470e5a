"""CPU-only checks of the product's host side: the C-ABI library loads, exports every symbol
include/nfisam_hip.h declares, and the kernel<->reference parameter layout map is a bijection.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import nfisam_hip as nh
from oracle import nsf_torch as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    nh.build()
    lib = nh.lib()
    hdr = open(os.path.join(ROOT, "include", "nfisam_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nfisam_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(nh.EXPORTS), declared ^ set(nh.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nfisam_abi_version() == 1600


def test_struct_sizes_match_header(tmp_path):
    # what a C compiler makes of include/nfisam_hip.h (the binding's ctypes mirrors must agree with it)
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "nfisam_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(nfisam_train_state), sizeof(nfisam_adam_cfg), sizeof(nfisam_clique), sizeof(nfisam_post_clique), '
                   'sizeof(nfisam_sim_op)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(nh.TrainState), C.sizeof(nh.AdamCfg), C.sizeof(nh.Clique), C.sizeof(nh.PostClique),
                     C.sizeof(nh.SimOp)], sizes
    assert C.sizeof(nh.TrainState) == 32
    assert C.sizeof(nh.AdamCfg) == 32
    assert C.sizeof(nh.Clique) == 64


@pytest.mark.parametrize("D,K,H", [(1, 9, 8), (2, 5, 8), (6, 9, 8), (11, 9, 8), (16, 12, 8), (30, 9, 8)])
def test_layout_map_is_a_bijection_onto_reference_order(D, K, H):
    P, Pk = nh.param_count(D, K, H), nh.kparam_count(D, K, H)
    assert P == O.param_count(D, K, H)       # D=11,K=9,H=8 -> 3606 etc.
    m = nh.layout_map(D, K, H)
    assert m.shape == (Pk,)
    real = m[m >= 0]
    assert real.size == P and np.array_equal(np.sort(real), np.arange(P))
    assert Pk % 4 == 0 and Pk >= P


def test_param_counts_from_survey():
    assert nh.param_count(11, 9, 8) == 3606
    assert nh.param_count(6, 9, 8) == 1716
    assert nh.param_count(30, 9, 8) == 12612


def test_pack_unpack_roundtrip_and_transposes():
    D, K, H, L = 5, 9, 8, 2
    P = nh.param_count(D, K, H)
    blob = torch.arange(L * P, dtype=torch.float32)
    kb = nh.pack(blob, D, K, H, L)
    assert kb.numel() == L * nh.kparam_count(D, K, H)
    assert torch.equal(nh.unpack(kb, D, K, H, L), blob)
    # spot-check: W2t[k][o] of dim i=2 equals W2[o][k]
    init, nets = O.unpack(blob[:P], D, K, H)
    Po = 3 * K - 1
    ND0 = K // 2
    HP = (K + ND0 + 3) // 4 * 4
    PoP = 2 * HP

    def col(o):     # reference output index -> kernel column (include/nfisam_hip.h)
        if o < K:
            return o
        if o < 2 * K:
            return HP + (o - K)
        j = o - 2 * K
        return K + j if j < ND0 else HP + K + (j - ND0)
    kfixed = H + H * H + H + H * PoP + PoP
    i = 2
    off = PoP + (i - 1) * kfixed + H * ((i - 1) * i // 2)
    oW2 = i * H + H + H * H + H
    W2 = nets[i - 1][4]
    for k in (0, 3, 7):
        for o in (0, 13, 25):
            assert kb[off + oW2 + k * PoP + col(o)] == W2[o, k]
    W0 = nets[i - 1][0]
    assert kb[off + 1 * H + 5] == W0[5, 1]


def test_compute_entry_points_refuse_cpu_tensors():
    x = torch.zeros(4, 3)
    kp = torch.zeros(nh.kparam_count(3, 9, 8))
    with pytest.raises(RuntimeError):
        nh.forward(x, kp, 9, 8, 5.0)
    with pytest.raises(RuntimeError):
        nh.inverse(x, None, kp, 9, 8, 5.0)


def test_supported_instantiations():
    assert nh.supported(9, 8) and nh.supported(5, 8) and nh.supported(12, 8)
    # every hidden_dim up to 16 (the reference takes any width, src/flows/flows.py:26-41): the next compiled width, zero-padded
    assert all(nh.supported(9, H) for H in range(1, 17)) and not nh.supported(9, 17) and not nh.supported(17, 8)


def test_uncompiled_hidden_widths_are_zero_padded_into_the_next_compiled_width():
    """hidden_dim 6 / 12 (the reference's own grid: example/slam/manhattan_world_with_range/lawnmower_4x4/run_nfisam.py:5-6)
    live in the kernel layout of width 8 / 16: same kernel-parameter count, the reference's parameter count and order, padding
    entries marked -1 (packed as exact zeros), pack / unpack a bijection on the real entries."""
    for H, Hc in ((1, 4), (3, 4), (6, 8), (7, 8), (10, 16), (12, 16)):
        D, K = 5, 9
        Po = 3 * K - 1
        P = Po + sum(i * H + H + H * H + H + H * Po + Po for i in range(1, D))
        assert nh.param_count(D, K, H) == P
        assert nh.kparam_count(D, K, H) == nh.kparam_count(D, K, Hc)
        m = nh.layout_map(D, K, H)
        real = m[m >= 0]
        assert real.size == P and np.array_equal(np.sort(real), np.arange(P))
        blob = torch.randn(P)
        kb = nh.pack(blob, D, K, H)
        assert kb.numel() == nh.kparam_count(D, K, Hc) and torch.equal(nh.unpack(kb, D, K, H), blob)
        assert torch.all(kb[torch.from_numpy(m < 0)] == 0)
        # the same model written as a width-Hc model with zero rows / columns packs to the same kernel blob
        wide = torch.zeros(nh.param_count(D, K, Hc))
        t, tw = Po, Po
        wide[:Po] = blob[:Po]
        for i in range(1, D):
            for (r, c, rw, cw) in ((H, i, Hc, i), (H, 1, Hc, 1), (H, H, Hc, Hc), (H, 1, Hc, 1), (Po, H, Po, Hc), (Po, 1, Po, 1)):
                blk = torch.zeros(rw, cw)
                blk[:r, :c] = blob[t:t + r * c].reshape(r, c)
                wide[tw:tw + rw * cw] = blk.reshape(-1)
                t += r * c; tw += rw * cw
        assert torch.equal(nh.pack(wide, D, K, Hc), kb)


def test_posterior_walk_refuses_a_table_that_would_read_past_the_latent_draws():
    """Zt holds ONE row per frontal column, consumed in walk order (include/nfisam_hip.h: nfisam_nsf_posterior_walk): a table
    whose frontal columns add up to more than total_dim, or with a column index outside the sample matrix, would make the
    kernel read / write out of bounds -- the binding raises before anything is sent to the device."""
    D, K, H = 5, 9, 8
    def entry(front, sep):
        return dict(kparams=torch.zeros(nh.kparam_count(D, K, H)), mean=torch.zeros(D), std=torch.ones(D),
                    circular=torch.zeros(D, dtype=torch.uint8), D_model=D, obs=np.zeros(0), sep_cols=sep, front_cols=front)
    with pytest.raises(ValueError, match="frontal columns exceed"):
        nh.posterior_walk([entry([0, 1, 2], []), entry([2, 3], [0, 1])], 4, 8, K, H, 5.0, 1, torch.device("cpu"))
    with pytest.raises(ValueError, match="out of range"):
        nh.posterior_walk([entry([0, 1, 2], []), entry([4], [0, 1])], 4, 8, K, H, 5.0, 1, torch.device("cpu"))
    table = np.zeros(2, dtype=nh.POST_DTYPE)
    table["n_frontal"] = [3, 2]
    with pytest.raises(ValueError, match="frontal columns exceed"):
        nh.posterior_walk_raw(table, np.array([0, 1, 2, 0, 1, 2, 3]), np.zeros(0), 4, 8, D, K, H, 5.0, 1, torch.device("cpu"))


def test_workspace_count_covers_groups_of_up_to_sixteen_blocks():
    """`nfisam_nsf_grad_workspace_count` (host-only): a one-layer clique's workspace must hold what the chunk-persistent kernel
    addresses in it -- gradient copies, loss ring + 64 counters, the second set of copies, the second (theta | m | v) buffer, two
    sets of TAGGED copies (2 floats per parameter) and the theta exchange: (6 x copies + 5) x kparam_count + ring + counters for
    `copies` = ceil(n / 256) four-wave blocks per (clique, dim) group, up to sixteen (n <= 4096, round 5); and it does not shrink
    with n."""
    lib = nh.lib()
    lib.nfisam_nsf_grad_workspace_count.restype = C.c_size_t
    K, H, L = 9, 8, 1
    prev = 0
    for n in (64, 256, 257, 2000, 2048, 2049, 3000, 4096):
        for D in (1, 6, 15):
            kc = nh.kparam_count(D, K, H)
            copies = (n + 255) // 256
            need = (6 * copies + 5) * kc + 128 * 128 + 64
            if (n + 127) // 128 <= 16:      # round 6: blocks of 128 particles (two lanes per particle, csrc/nsf_half.h), up to sixteen per group
                need = max(need, (6 * ((n + 127) // 128) + 5) * kc + 128 * 128 + 64)
            got = int(lib.nfisam_nsf_grad_workspace_count(n, D, K, H, L))
            assert got >= need, (n, D, got, need)
        now = int(lib.nfisam_nsf_grad_workspace_count(n, 6, K, H, L))
        assert now >= prev
        prev = now
