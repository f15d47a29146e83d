"""Turn rocprofv3's rocpd sqlite outputs (kernel trace + PMC passes) into the text summaries
kept under profiles/.  usage: summarize_rocprof.py <prof_dir> <out_prefix>"""
import glob
import json
import os
import sqlite3
import sys

prof, prefix = sys.argv[1], sys.argv[2]
lines = []
trace = os.path.join(prof, "trace", "trace_results.db")
if os.path.exists(trace):
    c = sqlite3.connect(trace).cursor()
    lines.append("# rocprofv3 --kernel-trace --stats : per-kernel totals (durations in ns)")
    lines.append("name,calls,total_ns,average_ns,percent")
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(",".join(str(x) for x in r))
    lines.append("")
    lines.append("# dispatches of the trace kernel")
    lines.append("name,duration_ns,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size")
    for r in c.execute("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size "
                       "from kernels where name like '%trace%' order by start"):
        lines.append(",".join(str(x) for x in r))
open(prefix + "_kernel_stats.csv", "w").write("\n".join(lines) + "\n")

pmc = {}
for db in sorted(glob.glob(os.path.join(prof, "pmc_*", "pmc_results.db"))):
    c = sqlite3.connect(db).cursor()
    q = ("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
         "group by kernel_name, dispatch_id, counter_name order by dispatch_id")
    for name, disp, counter, value in c.execute(q):
        if "trace" not in name:
            continue
        pmc.setdefault(counter, []).append(value)
rows = ["# rocprofv3 --pmc passes, rtow_trace_kernel, one value per dispatch (sum over XCDs/SEs)",
        "counter,per_dispatch_values"]
for k in sorted(pmc):
    rows.append(k + "," + " ".join(f"{v:.6g}" for v in pmc[k]))
open(prefix + "_pmc.csv", "w").write("\n".join(rows) + "\n")
print("\n".join(lines + rows))
