#!/usr/bin/env python3
"""Summarises rocprofv3 (rocpd sqlite) outputs: per-kernel trace stats and PMC counters.
usage: python profiles/summarize.py <trace.db> [pmc.db ...] > profiles/rNN_<name>.txt"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        print(f"== {path}")
        n = c.execute("select count(*) from kernels").fetchone()[0]
        has_pmc = c.execute("select count(*) from counters_collection").fetchone()[0]
        if n and not has_pmc:
            print("-- kernel trace (rocprofv3 --kernel-trace --stats): name, calls, avg/min/max ms, total ms, vgpr, agpr, lds")
            cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
            grid = next((x for x in ("grid_size_x", "grid_x", "grid_size") if x in cols), None)
            # one line per (kernel, grid size): the same kernel runs the timed step and the small latency / test launches
            q = ("select name, count(*), avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6, sum(end-start)/1e6,"
                 " max(vgpr_count), max(accum_vgpr_count), max(lds_size)" + (", " + grid if grid else ", 0") +
                 " from kernels group by name" + (", " + grid if grid else "") + " order by 6 desc")
            rows = list(c.execute(q))
            for r in rows:
                print("%-72s n=%-3d avg=%9.3f min=%9.3f max=%9.3f total=%9.1f vgpr=%s agpr=%s lds=%s grid=%s" % ((r[0][:72],) + r[1:]))
            # kernels whose launches differ widely under one (name, grid) - the timed step's launches beside small calls, or the gated
            # launches of the binary path that leave at once beside the ones that run: one line per duration cluster
            # (sorted durations, a new cluster wherever the next one is more than 1.5x the last)
            print("-- duration clusters of the kernels above whose launches differ widely (same name and grid): n, avg / min / max ms")
            for r in rows:
                if r[4] > 1.5 * r[3] and r[4] > 0.05:
                    cond = " and %s = %d" % (grid, r[9]) if grid else ""
                    ds = sorted(x[0] / 1e6 for x in c.execute("select end-start from kernels where name = ?" + cond, (r[0],)))
                    cl = [[ds[0]]]
                    for d in ds[1:]:
                        if d > 1.5 * cl[-1][-1]:
                            cl.append([])
                        cl[-1].append(d)
                    for g in cl:
                        print("%-72s n=%-3d avg=%9.3f min=%9.3f max=%9.3f grid=%s" % (r[0][:72], len(g), sum(g) / len(g), g[0], g[-1], r[9]))
        if has_pmc:
            # one line per (kernel, grid size, counter): the same kernel runs the timed step's launches beside the small online / test launches,
            # and gated launches that leave at once beside the ones that run - `per launch` is the average over the dispatches
            # above a tenth of the largest (`big` of them), `sum` / `n` are over all
            print("-- PMC (rocprofv3 --pmc ...): kernel, grid, counter, per launch (over `big` dispatches above a tenth of the largest), sum over all n dispatches")
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            grid = "grid_size" if "grid_size" in cols else "0"
            disp = next((x for x in ("dispatch_id", "correlation_id") if x in cols), None)
            per = {}
            has_t = "start" in cols and "end" in cols
            if disp:
                q = (f"select kernel_name, {grid}, counter_name, {disp}, sum(value)" + (", min(end - start)" if has_t else ", 0") + " from counters_collection"
                     f" group by kernel_name, {grid}, counter_name, {disp}")
                for name, g, ctr, d, v, dur in c.execute(q):
                    per.setdefault((name, g, ctr), []).append((dur or 0, v))
            else:
                for name, g, ctr, v, n_ in c.execute(f"select kernel_name, {grid}, counter_name, sum(value), count(*) from counters_collection group by 1, 2, 3"):
                    per[(name, g, ctr)] = [(0, v / n_)] * n_
            for (name, g, ctr), dv in sorted(per.items(), key=lambda kv: (kv[0][0], kv[0][1], kv[0][2])):
                # dispatches of one (kernel, grid) whose durations differ widely are different launches (one channel | both channels of the SC
                # matcher; gated launches that leave at once): one line per duration cluster (sorted, a new cluster where the next is > 1.5x),
                # labelled with the cluster's mean duration UNDER THE PROFILER (ms) - match it with the kernel trace's clusters by rank
                dv.sort()
                cl = [[dv[0]]]
                for x in dv[1:]:
                    if has_t and x[0] > 1.5 * cl[-1][-1][0] and x[0] > 20000:
                        cl.append([])
                    cl[-1].append(x)
                allv = [v for _, v in dv]
                for ci, gcl in enumerate(cl):
                    vals = [v for _, v in gcl]
                    big = [v for v in vals if v > 0.1 * max(vals)] or vals
                    tag = ("cluster %d/%d ~%.3f ms" % (ci + 1, len(cl), sum(d for d, _ in gcl) / len(gcl) / 1e6)) if len(cl) > 1 else "-"
                    print("%-72s grid=%-9s %-28s per launch %.6g (big=%d)  sum %.6g (n=%d)  [%s]" % (name[:72], g, ctr, sum(big) / len(big), len(big), sum(vals), len(vals), tag))


if __name__ == "__main__":
    main()
