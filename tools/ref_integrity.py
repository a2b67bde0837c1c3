"""Integrity gate of the golden-vector generators (tools/gen_golden_*.py): they exec line ranges of files under
/root/reference -- untrusted, read-only content.  Every file is hashed before a single line of it is executed; a file
that drifted from the reviewed revision aborts the generator.  Never on an automated path: the tests read only
the committed .npz fixtures."""
import hashlib
import os

REF = "/root/reference"

# sha256 of the reviewed revision of every reference file a generator reads
REVIEWED = {
    "configs/data_utils.py": "5cfb8ee27a56cf15b0654ec86f379e654a5ff3fab388dad454b4990089348098",
    "utils/metrics_ranking.py": "14bf409b36968ba898d27e4970a312f4c3994598af9f211c424086c42685b6d6",
    "utils/utilities.py": "3aeed80cecc6bb42274480f27101f5469b97c33ed11040ca9c9cf8cc6840e1d5",
    "utils/objectives.py": "bc26cc3417b4310b9262ba3646a25b04c7d5aaef1063b0ed8e0544b8f95ca0be",
    "modules/interaction/interaction_dot.py": "062bb990d91f75f8fb534d074ac45da3532d9b0c1900cb2c39b481e9390e0837",
    "sampler/nodesampler.cpp": "735aea5b59a48465fffa53885ec4419ab45bf0d0c576bf9079b63059b1a84fcf",
    "sampler/sampler.pyx": "fbaea37387c11e38295b95827ebdc2f609d3b7d8cf2fbf2fc66da1985b65afc4",
}


def read_reviewed(rel):
    """text of REF/rel after its sha256 matched the reviewed revision; raises otherwise"""
    path = os.path.join(REF, rel)
    data = open(path, "rb").read()
    got = hashlib.sha256(data).hexdigest()
    want = REVIEWED.get(rel)
    if want is None:
        raise SystemExit("ref_integrity: %s is not on the reviewed list" % rel)
    if got != want:
        raise SystemExit("ref_integrity: %s changed (sha256 %s, reviewed %s): review it before generating fixtures from it"
                         % (rel, got, want))
    return data.decode("utf-8")


def lines(rel):
    return read_reviewed(rel).split("\n")
