#!/usr/bin/env python3
"""Turn rocprofv3 (rocpd .db) outputs into the small summaries kept under profiles/.
usage: profile_extract.py stats <db> <out.csv>
       profile_extract.py stats_by_grid <db> <out.csv> [min grid]   (one line per kernel AND grid size: separates
                                                                     the 4,096-block launches from the small ones)
       profile_extract.py traffic <write_db> <fetch_db> <out.json> <kernel substring> <algorithmic bytes per launch> [commit]
       profile_extract.py sq <out.json> <db> [<db> ...]        (every counter of the pmc passes, averaged per kernel and grid size)"""
import csv, json, sqlite3, sys


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5]])


def stats_by_grid(db, out, min_grid=0):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels where grid_x >= ? group by name, grid_x order by 4 desc", (min_grid,)).fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], round(r[4], 1), r[5], r[6]])


def counter(db, name, kernel):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, sum(value), max(grid_size), max(lds_block_size), max(vgpr_count), max(kernel_name) "
                       "from counters_collection where counter_name=? and kernel_name like ? group by dispatch_id",
                       (name, "%" + kernel + "%")).fetchall()
    big = max(r[2] for r in rows)
    vals = [r[1] for r in rows if r[2] == big]        # the full-size launches only
    return dict(n=len(vals), mean=sum(vals) / len(vals), min=min(vals), max=max(vals)), rows[0][5], rows[0][3], rows[0][4], big


def sq(out, dbs):
    res = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, grid_size, counter_name, dispatch_id, sum(value) from counters_collection "
                           "group by kernel_name, grid_size, counter_name, dispatch_id").fetchall()
        acc = {}
        for name, grid, ctr, disp, val in rows:
            acc.setdefault(("%s grid=%d" % (name.split("(")[0], grid), ctr), []).append(val)
        for (k, ctr), vals in acc.items():
            d = res.setdefault(k, {})
            d[ctr] = sum(vals) / len(vals)
            d["launches"] = max(d.get("launches", 0), len(vals))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


def traffic(wdb, rdb, out, kernel, alg, commit=None):
    w, name, lds, vgpr, grid = counter(wdb, "WRITE_SIZE", kernel)
    r, _, _, _, _ = counter(rdb, "FETCH_SIZE", kernel)
    wb, rb = w["mean"] * 1024, r["mean"] * 1024 * 2
    json.dump({
        "commit": commit, "kernel": name.split("(")[0],
        "command": "rocprofv3 --kernel-trace --pmc <COUNTER> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra (one pass per counter)",
        "note": "MI355X_MICROARCH.md HBM section: WRITE_SIZE is exact for 16-B/lane streaming stores; FETCH_SIZE reads 1/2 of wide coalesced reads on gfx950 -> doubled. Units are KiB.",
        "kernel_name_seen": name, "lds_block_size": lds, "vgpr": vgpr, "grid": grid,
        "WRITE_SIZE_KiB_per_launch": w, "FETCH_SIZE_KiB_per_launch": r,
        "hbm_write_bytes_per_launch": wb, "hbm_read_bytes_per_launch_corrected_x2": rb,
        "hbm_traffic_bytes_per_launch": wb + rb, "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (wb + rb) / alg}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats_by_grid":
        stats_by_grid(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3:])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), sys.argv[7] if len(sys.argv) > 7 else None)
