"""Turn rocprofv3's rocpd sqlite outputs (kernel trace + PMC passes) into the text summaries
kept under profiles/.  usage: summarize_rocprof.py <prof_dir> <out_prefix>"""
import glob
import json
import os
import sqlite3
import sys

prof, prefix = sys.argv[1], sys.argv[2]
lines = []
alone = []
trace = os.path.join(prof, "trace", "trace_results.db")
if os.path.exists(trace):
    c = sqlite3.connect(trace).cursor()
    lines.append("# rocprofv3 --kernel-trace --stats : per-kernel totals (durations in ns)")
    lines.append("name,calls,total_ns,average_ns,percent")
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(",".join(str(x) for x in r))
    lines.append("")
    lines.append("# dispatches of the trace kernel, in start order.  bench.py keeps two steps in flight, so launches of")
    lines.append("# the timed region overlap their neighbours (overlap_ns > 0) and stretch each other; the last three launches")
    lines.append("# run one at a time and are the ones roofline.kernel_ms is taken from")
    lines.append("name,start_ns,duration_ns,overlap_with_previous_ns,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size")
    rows = list(c.execute("select name,start,end,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size "
                          "from kernels where name like '%trace%' order by start"))
    t0 = rows[0][1] if rows else 0
    main = rows[0][0] if rows else ""   # the instance the timed region runs; bench.py ends with a few launches of the other instances
    prev_end = None
    others = {}
    for k, r in enumerate(rows):
        ov = max(0, prev_end - r[1]) if prev_end is not None else 0
        nxt = rows[k + 1][1] if k + 1 < len(rows) else None
        if ov == 0 and (nxt is None or nxt >= r[2]):
            (alone if r[0] == main else others.setdefault(r[0], [])).append(r[3])
        prev_end = r[2]
        lines.append(",".join(str(x) for x in (r[0], r[1] - t0, r[3], ov) + tuple(r[4:])))
    if alone:
        lines.append(f"# launches of {main} that ran alone: {len(alone)}, average duration {sum(alone) / len(alone):.0f} ns")
    for name, d in others.items():
        lines.append(f"# launches of {name} that ran alone: {len(d)}, average duration {sum(d) / len(d):.0f} ns")
open(prefix + "_kernel_stats.csv", "w").write("\n".join(lines) + "\n")

pmc = {}
pmc_other = {}
main_pmc = None
for db in sorted(glob.glob(os.path.join(prof, "pmc_*", "pmc_results.db"))):
    c = sqlite3.connect(db).cursor()
    q = ("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
         "group by kernel_name, dispatch_id, counter_name order by dispatch_id")
    for name, disp, counter, value in c.execute(q):
        if "trace" not in name:
            continue
        if main_pmc is None:
            main_pmc = name           # the first trace dispatch of a pass: the instance the timed region runs
        if name != main_pmc:          # the other instances bench.py launches at the end: no first test, slab test, bundle test
            pmc_other.setdefault(name, {}).setdefault(counter, []).append(value)
            continue
        pmc.setdefault(counter, []).append(value)
rows = [f"# rocprofv3 --pmc passes, {main_pmc}, one value per dispatch (sum over XCDs/SEs)",
        "counter,per_dispatch_values"]
for k in sorted(pmc):
    rows.append(k + "," + " ".join(f"{v:.6g}" for v in pmc[k]))
# the last template argument of rtow_trace_kernel says what the camera rays see before the sphere screen
labels = {"-1": "nothing in front of the screen (RTOW_NO_SLAB_TEST)", "0": "slab test along x", "1": "slab test along y",
          "2": "slab test along z", "3": "bundle test (RTOW_CAMERA_SCREEN=bundle: packet culling, not part of any quoted figure)",
          "4": "interval test alone (RTOW_CAMERA_SCREEN=interval)", "5": "mask test, then interval test (the default)"}
def label(name):
    import re
    m = re.search(r"rtow_trace_kernel<[^>]*,\s*(-?\d+)>", name)
    return labels.get(m.group(1), name) if m else name
rows[0] += f"  [{label(main_pmc or '')}]"
for other in pmc_other:
    rows.append("")
    rows.append(f"# the same counters for the launches of {other}  [{label(other)}] at the end of bench.py")
    for k in sorted(pmc_other[other]):
        rows.append(k + "," + " ".join(f"{v:.6g}" for v in pmc_other[other][k]))
if pmc_other:
    rows.append("")
# figures derived from the LAST dispatch of each pass (a launch that ran alone).  SQ counters count quad-cycles
# summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
def last(name):
    return pmc[name][-1] if name in pmc else None


need = ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_WAVES")
if all(last(n) is not None for n in need):
    cycles = last("GRBM_GUI_ACTIVE") / 8.0                      # shader clocks the launch lasted
    simd_cycles = cycles * 1024.0
    rows.append("")
    rows.append("# derived (last dispatch of the passes)")
    rows.append(f"kernel_cycles,{cycles:.6g}")
    if alone:   # the un-overlapped launches of the kernel-trace pass of the same command on the same box
        ms = min(alone) * 1e-6
        ghz = cycles / (ms * 1e6)
        rows.append(f"shader_clock_ghz,{ghz:.3f},# kernel_cycles / {ms:.2f} ms (shortest un-overlapped launch of the trace pass); "
                    f"the 157.3 TFLOP/s peak assumes 2.4 GHz: at this clock it is {157.3 * ghz / 2.4:.1f}")
    rows.append(f"occupancy_waves_per_simd,{4.0 * last('SQ_WAVE_CYCLES') / simd_cycles:.3f},# 4 x SQ_WAVE_CYCLES / (1024 SIMDs x cycles); waves launched: {last('SQ_WAVES'):.0f}")
    rows.append(f"valu_busy_fraction,{4.0 * last('SQ_ACTIVE_INST_VALU') / simd_cycles:.3f},# share of SIMD cycles in which a VALU instruction executes")
    rows.append(f"valu_lane_utilisation,{last('SQ_THREAD_CYCLES_VALU') / (64.0 * last('SQ_ACTIVE_INST_VALU')):.3f},# active lanes per executing VALU instruction / 64 (1 - divergence)")
    rows.append(f"valu_cycles_per_instruction,{4.0 * last('SQ_ACTIVE_INST_VALU') / last('SQ_INSTS_VALU'):.2f}")
    f64 = sum(last(n) or 0 for n in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
    if f64:
        rows.append(f"valu_f64_share_of_instructions,{f64 / last('SQ_INSTS_VALU'):.4f},# fma + add + mul in binary64 (primary segments)")
open(prefix + "_pmc.csv", "w").write("\n".join(rows) + "\n")
print("\n".join(lines + rows))
