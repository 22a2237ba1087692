"""Derived figures per kernel from scripts/pmc_summary.py's output (sums of rocprofv3 counters over a frame)."""
import re, sys
cur, data = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+)\s+dispatches=(\d+)", line)
    if m:
        cur = m.group(1); data[cur] = {"dispatches": int(m.group(2))}
    elif cur and line.strip():
        k, v = line.split()
        data[cur][k] = float(v)
for k, d in data.items():
    if not k.startswith("k_") or "TCP_GATE_EN1_sum" not in d or d.get("SQ_WAVE_CYCLES", 0) == 0 or "TCP_TCC_READ_REQ_sum" not in d:
        continue
    g = d["TCP_GATE_EN1_sum"]
    print(f"{k} ({d['dispatches']} launches)")
    print(f"   vector L1 hit rate            {1 - d['TCP_TCC_READ_REQ_sum'] / max(1, d['TCP_TOTAL_CACHE_ACCESSES_sum']):.3f}   (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)")
    print(f"   texture-data path busy        {d['TD_TD_BUSY_sum'] / g:.3f}   of which stalled on L1/L2 {d['TD_TC_STALL_sum'] / max(1, d['TD_TD_BUSY_sum']):.3f}")
    print(f"   texture-address path busy     {d['TA_TA_BUSY_sum'] / g:.3f}")
    print(f"   waves waiting (any counter)   {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f}   waiting to issue {d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:.3f}")
    # SQ_BUSY_CYCLES is summed over the 32 shader arrays; 1024 SIMDs.  A wave64 VALU instruction occupies a SIMD-32 for
    # 2 cycles (f32 add/mul/fma; f64, integer multiplies and transcendentals take 2-4x as long), and ONE wave can
    # issue one every 4 cycles: the first figure is the share of the SIMDs' issue capacity (a lower bound on how
    # busy the ALUs are), the second the share of the time in which some wave has a vector instruction in flight.
    simd_cycles = 32 * max(1, d['SQ_BUSY_CYCLES'])
    print(f"   vector ALU issue capacity     {2 * d['SQ_INSTS_VALU'] / simd_cycles:.3f}   (2 cycles x SQ_INSTS_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32); lower bound)")
    print(f"   vector instruction in flight  {4 * d.get('SQ_ACTIVE_INST_VALU', d['SQ_INSTS_VALU']) / simd_cycles:.3f}   (4 cycles x SQ_ACTIVE_INST_VALU, summed over waves, per SIMD cycle)")
    print(f"   instructions: VALU {d['SQ_INSTS_VALU'] / 1e9:.2f} G  SALU {d['SQ_INSTS_SALU'] / 1e9:.2f} G  VMEM read {d['SQ_INSTS_VMEM_RD'] / 1e9:.3f} G  LDS {d.get('SQ_INSTS_LDS', 0) / 1e9:.3f} G  branch {d['SQ_INSTS_BRANCH'] / 1e9:.2f} G")
