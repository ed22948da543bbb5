"""BASELINE config 3 on ONE GPU with more users than the benched 10 M — past 2^31 raw-log rows (~20.5 M users) — on the default
path, a sample of its users replayed by the oracle (tests/oracle_spot_check.py): VERDICT round 5, item 3.  218 GB of workspace +
log at 21 M users, 52 GB more for the sorted copy: the population DESIGN.md §3 quotes as the limit of one 288 GB GPU.
    python tools/beyond_2_31_run.py [users] [--out FILE]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import oracle_spot_check as osc  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 21_000_000
out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else ''
res = osc.spot_check('c3', users, 2000, p_click_modes=(False,))
for line in res:
    c = line['counters']
    line['real_rows'] = c['organic'] + c['bandit']
    line['raw_rows'] = c['log_rows']              # incl. the unused entries of the walk's reserved row chunks
    line['beyond_2_31_raw_rows'] = c['log_rows'] > (1 << 31) and c['log_dropped'] == 0
    line['beyond_2_31_real_rows'] = line['real_rows'] > (1 << 31)
    line['events_per_s_incl_reset'] = None
    s = json.dumps(line)
    print(s)
    if out:
        open(out, 'a').write(s + '\n')
    assert line['beyond_2_31_raw_rows'], 'the run stayed below 2^31 rows'
