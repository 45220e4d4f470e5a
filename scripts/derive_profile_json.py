"""Turn the per-launch PMC means of scripts/collect_profiles.sh (<dir>/pmc_means.json) into the two derived summaries
under profiles/: issue utilisation of the training kernel (C3 and the 64-clique batch) and its HBM-side traffic.
usage: python scripts/derive_profile_json.py gpurun_out/<dir> profiles/r02"""
import json, sys

src, dst = sys.argv[1], sys.argv[2]
pm = json.load(open(src + "/pmc_means.json"))
SIMDS, SES = 1024, 32


def kernel_of(d):
    """the training kernel of the run: the chunk-persistent instantiation (<K, H, true>) when the plan took it (its launches
    run a whole chunk of iterations), else the one-launch-per-iteration kernel"""
    ks = [k for k in d if "nsf_train1_" in k]
    per = [k for k in ks if "true" in k]
    name = (per or ks)[0]
    return dict(d[name], _name=name, _persistent=bool(per))


def iterations_per_launch(src, key):
    """bench.py's own line of the profiled run (collect_profiles.sh keeps it): iterations a launch of the training kernel runs"""
    try:
        line = json.load(open(src + "/%s_bench_line.json" % key))
        return int(line["roofline"].get("iterations_per_launch", 1))
    except Exception:   # noqa: BLE001
        return 1


out = {"_formulas": {
    "kernel_cycles": "SQ_BUSY_CYCLES / 32 (the counter sums the 32 shader engines' busy cycles)",
    "valu_issue_frac": "2 * SQ_ACTIVE_INST_VALU / (kernel_cycles * 1024 SIMDs): the counter advances 1 per plain VALU instruction "
                       "and 2 per transcendental (scripts/exp/valu_rate.hip); 2 cycles per plain fp32 VALU instruction is the BEST case "
                       "of the issue port -- under load the kernel's phases measure ~2.9 cycles per instruction (scripts/stamps3.py, "
                       "DESIGN.md 3.1b), so this fraction understates how busy the port is",
    "instruction_counts": "SQ_INSTS_VALU includes the MFMA instructions (static count of the tile loop: ~560 VALU + 212 MFMA per unit)",
    "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024)",
    "issue_frac": "valu_issue_frac + mfma_busy_frac (MFMA and VALU issue of a SIMD do not overlap: profiles/history/r02_mfma_valu_issue_microbench.txt)",
    "scalar_cache_hit_rate": "SQC_DCACHE_HITS / (SQC_DCACHE_HITS + SQC_DCACHE_MISSES)"}}
for name, key in (("C3", "c3"), ("batch64", "b64")):
    k = kernel_of(pm[key])
    ipl = iterations_per_launch(src, key) if k["_persistent"] else 1
    cyc = k["SQ_BUSY_CYCLES"] / SES
    valu = 2.0 * k["SQ_ACTIVE_INST_VALU"] / (cyc * SIMDS)
    mfma = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS)
    out[name] = {
        "kernel": "nsf_train1_kernel<9,8,true>" if k["_persistent"] else "nsf_train1_plain_kernel<9,8>", "iterations_per_launch": ipl,
        "kernel_cycles": cyc, "kernel_cycles_per_iteration": cyc / ipl, "waves_launched": k["SQ_WAVES"],
        "valu_instructions_per_iteration": k["SQ_INSTS_VALU"] / ipl, "mfma_instructions_per_iteration": k["SQ_INSTS_MFMA"] / ipl,
        "valu_instructions": k["SQ_INSTS_VALU"], "mfma_instructions": k["SQ_INSTS_MFMA"], "lds_instructions": k["SQ_INSTS_LDS"],
        "salu_instructions": k["SQ_INSTS_SALU"], "smem_instructions": k["SQ_INSTS_SMEM"],
        "valu_issue_frac": valu, "mfma_busy_frac": mfma, "issue_frac": valu + mfma,
        "wait_inst_any_over_wave_cycles": k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_over_lds_active": k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_LDS_IDX_ACTIVE"], 1.0),
        "scalar_cache_hit_rate": k["SQC_DCACHE_HITS"] / (k["SQC_DCACHE_HITS"] + k["SQC_DCACHE_MISSES"]),
    }
json.dump(out, open(dst + "_issue_utilisation.json", "w"), indent=1)

k = kernel_of(pm["c3"])
ipl = iterations_per_launch(src, "c3") if k["_persistent"] else 1
P = [None]
alg = 723552                                         # bench.py: 608 KB (x) + 112 KB (parameters) for the C3 launch
# C3 parameter counts (K = 9, H = 8): count(D) = 32 + (D-1)*368 + 8*(D-1)*D/2
cnt = lambda D: 32 + (D - 1) * 368 + 8 * ((D - 1) * D // 2)
Pc = [cnt(D) for D in (6, 8, 8, 10, 10, 12, 12, 12)]
copies = 8
grad_bytes = copies * sum(Pc) * 4
state_bytes = 3 * sum(Pc) * 4
traffic = {
    "kernel": "nsf_train1_kernel<9,8,true>" if k["_persistent"] else "nsf_train1_plain_kernel<9,8>", "iterations_per_launch": ipl,
    "profiled_at": __import__("datetime").datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC"),      # (bench.py quotes it next to `roofline.traffic`)
    "workload": "bench.py headline (C3: 8 cliques, n=2000, D=6..12), training iterations "
              "(the launch also applies the previous iteration's Adam update)",
    "launches_averaged": int(k.get("_n", 0)) or None,
    "FETCH_SIZE_KB_raw": k["FETCH_SIZE"], "WRITE_SIZE_KB_raw": k["WRITE_SIZE"],
    "hbm_side_bytes_per_launch_lower": int(1024 * (k["FETCH_SIZE"] + k["WRITE_SIZE"])),
    "hbm_side_bytes_per_launch_upper": int(1024 * (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"])),
    "algorithmic_bytes_per_launch": alg,
    "bytes_the_launch_must_move": {"gradient_copies_written": grad_bytes, "adam_state_written": state_bytes,
                                   "gradient_copies_and_state_read_per_dim_block": "8 blocks per (clique, dim) each read the 8 "
                                   "copies + theta, m, v of the dim: %d B in total, L2 / Infinity-Cache hits after the first block" % (8 * (grad_bytes + state_bytes))},
    "note": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (scripts/collect_profiles.sh), means over the "
            "launches of the run (training iterations issued as ONE launch each: NFISAM_CHAINS=1).  The counters sit at the L2-fabric boundary and include Infinity-Cache hits.  FETCH_SIZE counts "
            "64 B per 128-B request for wide streaming reads on gfx950 (MI355X_MICROARCH.md), i.e. up to x2 (both bounds given).  "
            "WRITE_SIZE: one gradient copy per BLOCK (8 per clique, %.2f MB) on every launch + the new theta/m/v (%.2f MB) on the "
            "training launches that carry a pending Adam update (most of them: the kernel-timing loop launches the "
            "gradient alone): %.2f MB counted.  (With 8 VGPRs spilled the same counter read 6.2 MB: 2 KB of scratch per wave.)  "
            "FETCH_SIZE is dominated by the fused update: the 8 blocks of a (clique, dim) each read its 8 copies + theta, m, v "
            "(L2 / Infinity-Cache hits after the first).  At ~300 GB/s of fabric traffic the kernel is nowhere near the HBM "
            "roofline; the figure that matters is the issue utilisation (*_issue_utilisation.json)." % (grad_bytes / 1e6, state_bytes / 1e6, k["WRITE_SIZE"] * 1024 / 1e6),
}
traffic["hbm_side_bytes_per_iteration_lower"] = traffic["hbm_side_bytes_per_launch_lower"] / ipl
traffic["hbm_side_bytes_per_iteration_upper"] = traffic["hbm_side_bytes_per_launch_upper"] / ipl
if k["_persistent"]:
    # algorithmic bytes per sample-iteration = 4 D (x) + the parameters once per iteration: 723552 B per C3
    # iteration, x units of one launch = its iterations (the persistent form itself reads x only once per launch: the tiles
    # stay in LDS -- it needs fewer bytes than the survey's figure, which is kept as the yardstick)
    alg = 723552 * ipl
    traffic["algorithmic_bytes_per_launch"] = alg
    traffic["algorithmic_bytes_per_iteration"] = 723552
    traffic["note"] = ("chunk-persistent launch of %d iterations (NFISAM_CHAINS=1: one launch per chunk); separate rocprofv3 --pmc FETCH_SIZE / "
                       "--pmc WRITE_SIZE passes, means over the launches of the run.  The counters sit at the L2-fabric boundary and "
                       "include Infinity-Cache hits; FETCH_SIZE counts 64 B per 128-B request for wide streaming reads on gfx950 "
                       "(both bounds given).  WRITE side: per iteration every block sends its gradient copy out as (value, tag) pairs "
                       "with agent-scope WRITE-THROUGH stores (2 x %.2f MB: the tags double the bytes, and write-through means every "
                       "store reaches the fabric instead of staying in L2 -- the price of an exchange that is correct from any XCD), "
                       "the blocks publish the updated parameters the same way (2 x %.2f MB) and record theta, m, v (%.2f MB).  FETCH "
                       "side: a parameter's 8 copies are read by ONE thread of the group (update divided among the blocks), every "
                       "block reads the published parameters; L2 hits when the group sits on one XCD.  ~0.4 TB/s in all: latency, not "
                       "bandwidth, is what the exchange costs (DESIGN.md 3.1f)." % (ipl, grad_bytes / 1e6, sum(Pc) * 4 / 1e6, state_bytes / 1e6))
traffic["ratio_to_algorithmic"] = [traffic["hbm_side_bytes_per_launch_lower"] / alg, traffic["hbm_side_bytes_per_launch_upper"] / alg]
json.dump(traffic, open(dst + "_train_kernel_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)); print(json.dumps(traffic, indent=1)[:1500])
