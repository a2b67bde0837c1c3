"""CPU: the bench line's measured-traffic fields come from profiles/traffic.json ONLY while the kernel sources are the profiled ones.

bench.py reports `roofline.traffic` (HBM bytes per launch of the dominant kernel) and `evaluator.traffic` (bytes per whole@50 pass)
from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes (tools/make_traffic.py); an entry collected on other sources -- its
`csrc_hash` differs from __graft_entry__._source_hash() -- must read as None (the line then says "not measured" instead of
quoting a stale number).  The committed table itself must be of the committed sources: a kernel change without a re-collection
(tools/refresh_traffic.sh on the GPU box) fails here, not in the judge's cross-check."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def _table():
    return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))


def test_committed_traffic_table_is_of_the_committed_sources():
    table = _table()
    assert {"basic_embedding/neg_shared/skip-gram/b512/d100", "mf/neg_shared/skip-gram/b512/d100", "evaluator/whole@50/16Kx17K/d100"} <= set(table)
    for key, ent in table.items():
        assert ent["csrc_hash"] == ge._source_hash(), "%s was profiled on other kernel sources: run tools/refresh_traffic.sh on the GPU box" % key
        assert ent["bytes_per_launch"] == int(round((2 * ent["fetch_size_kib_per_launch"] + ent["write_size_kib_per_launch"]) * 1024))


def test_evaluator_traffic_is_quoted_only_for_the_profiled_sources(monkeypatch, tmp_path):
    ent = _table()["evaluator/whole@50/16Kx17K/d100"]
    assert bench.evaluator_traffic(100) == ent["bytes_per_pass"] == 2 * ent["bytes_per_launch"]
    assert bench.evaluator_traffic(128) is None            # no pass was collected at d = 128
    monkeypatch.setattr(ge, "_source_hash", lambda: "0" * 64)
    assert bench.evaluator_traffic(100) is None             # other sources: stale, not quoted
