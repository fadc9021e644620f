"""The one stdout line of bench.py: small, prose-free, parseable -- whatever the full result holds.  (Round 4's
22 KB line was printed and not parsed by the driver: the round counted as unmeasured.)"""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_result():
    """A real full result: the builder-run record of round 4 (21.9 KB, prose in every row)."""
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_final.json')) as fh:
        return json.load(fh)


def test_stdout_line_is_small_and_round_trips():
    full = _full_result()
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.MAX_LINE_BYTES, len(text)
    assert '\n' not in text
    back = json.loads(text)
    # the driver's contract fields survive unchanged
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data'):
        assert back[k] == full[k], k
    assert back['config']['workload'] == full['config']['workload']
    r = back['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert r[k] == full['roofline'][k], k
    assert len(r['traffic_source']) <= 40
    cb = back['cpu_baseline']
    assert cb['kind'] == 'reference' and cb['cores'] == full['cpu_baseline']['cores'] and cb['value'] > 0
    assert back['parity']['ok'] is True
    # one small object per secondary row, no strings longer than a label
    assert len(back['secondary']) == len(full['secondary'])
    for row in back['secondary']:
        assert len(json.dumps(row)) < 400, row
        assert row['ok'] is True
    assert max(len(v) for v in _strings(back)) <= 100


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_headline_survives_an_oversized_explanatory_part():
    full = _full_result()
    full['secondary'] = full['secondary'] * 40  # pathological: hundreds of rows
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.MAX_LINE_BYTES
    assert line['value'] == full['value'] and 'roofline' in line and 'cpu_baseline' in line


def test_failed_secondary_row_stays_small():
    row = bench.compact_row(dict(config='c4', error='RuntimeError: ' + 'x' * 5000))
    assert len(json.dumps(row)) < 200


def test_bare_multi_gpu_line():
    bare = dict(metric='SpMM GEdges/s', value=1.0, unit='GEdges/s', n_gpus=8, steps=20, warmup=5, ms_per_step=9.9,
                higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                config=dict(workload='w', reduce='sum', parallelism='row-sharded x8, exchange allgather'),
                extras='withheld: ' + 'y' * 1000)
    line = bench.compact_line(bare)
    assert line['n_gpus'] == 8 and len(json.dumps(line)) < 1000
