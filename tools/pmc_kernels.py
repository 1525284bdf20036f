#!/usr/bin/env python3
"""Hardware counters of several kernels of one bench configuration, each set in its own `rocprofv3 --pmc` pass (no tracing):
    python tools/pmc_kernels.py OUT.json [bench args]
Per kernel and launch: wave-cycles, VALU issue, waiting, LDS instructions / busy cycles / bank conflicts, matrix-core busy cycles,
HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KB; the gfx950 correction of the guide).  Uses bench.pmc_counters (child runs of bench.py)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

KERNELS = {"local-energy sensitivities + fused finish (two-wave matrix-core kernel)": "ff_eloc_mfma_kernel<6, 2, true, 2>",
           "local-energy sensitivities + finish (heavy route: the highest cost classes, one walker per wave)": "ff_wide_eloc_kernel<2, 1, true, double, true>",
           "theta-gradient adjoint (two waves per workgroup)": "ff_ode_adjtab_kernel<6, 2, 2>",
           "Metropolis sampler (beside the adjoint)": "ff_mcmc_spin_philox_kernel<3>",
           "flow": "ff_ode_fwd_kernel<6, 2, 0, true>"}
PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
          ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_INSTS_VALU"),
          ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS"),
          ("SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_SALU"))


def main():
    out_path, argv = sys.argv[1], sys.argv[2:]
    res = {}
    for label, name in KERNELS.items():
        c, why = bench.pmc_counters(name, argv, PASSES)
        if c is None:
            res[label] = {"kernel": name, "error": why}
            continue
        d = {"kernel": name, "per_launch": c}
        if c.get("SQ_WAVE_CYCLES"):
            d["valu_active"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
            d["wait_any"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        res[label] = d
        print(label, json.dumps({k: v for k, v in d.items() if k != "per_launch"}), flush=True)
    json.dump({"bench_args": argv, "kernels": res}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
