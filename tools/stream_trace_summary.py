#!/usr/bin/env python3
"""Which resource serialises the step kernels of several handles on one GPU?  Reads the rocpd SQLite file of
`rocprofv3 --kernel-trace -- python3 bench.py --streams S ...` and reports, for the step-kernel dispatches: the HIP stream ->
hardware queue mapping, how many run concurrently (share of the traced span), and the per-launch duration.

    python tools/stream_trace_summary.py gpurun_out/.../x_results.db [label]  > profiles/r03_streams4_trace.json
"""
import json, sqlite3, sys
from collections import Counter
import numpy as np

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select start, end, duration, stream_id, queue_id, grid_x from kernels where name like '%brs_step_kernel%' order by start"))
tail = rows[len(rows) // 2:]                      # steady state: the second half
ev = sorted([(r[0], 1) for r in tail] + [(r[1], -1) for r in tail])
cur, last, hist = 0, ev[0][0], Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
streams = sorted({r[3] for r in rows})
per_round = len(streams)
span_ms = (tail[-1][1] - tail[0][0]) / 1e6
out = dict(label=sys.argv[2] if len(sys.argv) > 2 else "", step_kernel_dispatches=len(rows), envs_per_launch=int(rows[0][5]),
           stream_to_hw_queue={str(s): sorted({r[4] for r in rows if r[3] == s}) for s in streams},
           distinct_hw_queues=len({r[4] for r in rows}),
           launch_ms_mean=float(np.mean([r[2] for r in tail]) / 1e6),
           concurrency_share_of_span={str(k): round(v / tot, 4) for k, v in sorted(hist.items())},
           ms_per_round_of_all_streams=span_ms / (len(tail) / per_round),
           finding="HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them the null stream's): streams that "
                   "share a queue run their kernels one after the other, and a launch of N/S envs lasts as long as one of N (a launch "
                   "lasts as long as its slowest wave)")
print(json.dumps(out, indent=1))
