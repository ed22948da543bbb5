"""Per-dispatch view of a rocprofv3 kernel trace of one run in rounds: for the named kernels, the duration of every launch in
launch order (which rounds carry a kernel's time: the few large ones or the many small ones).
    python tools/trace_rounds.py <kernel_trace.csv> <out.json> k_pick k_draw_tp k_advance_run ..."""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = sys.argv[3:]
out = {}
for nm in names:
    sel = [r for r in rows if re.search(nm, r['Kernel_Name'])]
    sel.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in sel]        # us
    grid = [int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) for r in sel]
    tot = sum(dur)
    srt = sorted(dur, reverse=True)
    out[nm] = dict(launches=len(dur), total_ms=round(tot / 1e3, 3),
                   top10_share=round(sum(srt[:10]) / max(tot, 1e-9), 3), median_us=round(srt[len(srt) // 2], 1) if srt else None,
                   below_100us=sum(1 for d in dur if d < 100), time_below_100us_ms=round(sum(d for d in dur if d < 100) / 1e3, 3),
                   first_40_us=[round(d, 1) for d in dur[:40]], first_40_grid=grid[:40],
                   every_10th_us=[round(d, 1) for d in dur[::10]])
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k, v in out.items():
    print(k, {a: b for a, b in v.items() if not isinstance(b, list)})
