"""Probe: where the matrix-core local-energy kernel's waves wait -- issue stalls, instruction fetch, instruction-cache misses
(each counter set in its own rocprofv3 --pmc pass of bench.py).   python tools/probes/pmc_ifetch.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
PASSES = (("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"),
          ("SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_WAVES", "SQ_INSTS_SMEM"),
          ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE"),
          ("SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_LDS"),
          ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA"),
          ("SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_BRANCH", "SQ_INSTS_CBRANCH_TAKEN"))
for name in ("ff_eloc_mfma_kernel<6, 2, true, 2>", "ff_ode_adjtab_kernel<6, 2, 2>"):
    c, why = bench.pmc_counters(name, [], PASSES)
    print(name, why, json.dumps(c, indent=1), flush=True)
