#!/usr/bin/env python3
"""Developer check: the epoch-parity case (tests/epoch_parity.run_case) with every whole@50 evaluation of the product run through
BOTH the one-kernel evaluator and predict + topk; prints the users whose metrics differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from nncf_amd import objectives, synthetic
    from tests import epoch_parity as ep
    scheme = sys.argv[1] if len(sys.argv) > 1 else "group_neg_shared"
    loss = sys.argv[2] if len(sys.argv) > 2 else "log-loss"
    orig = objectives.Evaluator._whole

    def both(self, model_predict, which, topk, user_chunk=2048, fused=None):
        a = orig(self, model_predict, which, topk, user_chunk, True)
        pe = getattr(model_predict, "predict_engine", None)
        st = self._device_state(pe() if pe is not None else model_predict.engine)
        ma, va = st["metrics"].clone().view(-1, 3).cpu().numpy(), st["valid"].clone().cpu().numpy()
        b = orig(self, model_predict, which, topk, user_chunk, False)
        mb, vb = st["metrics"].view(-1, 3).cpu().numpy(), st["valid"].cpu().numpy()
        bad = np.nonzero((va != vb) | ((ma != mb).any(axis=1) & (va != 0)))[0]
        sys.stderr.write("%s: fused recall %.9f two-kernel %.9f; users that differ: %d %s\n" % (which, a["recall@%d" % topk], b["recall@%d" % topk], bad.size, bad[:12]))
        for u in bad[:6]:
            sys.stderr.write("   user %d fused %s two-kernel %s\n" % (u, ma[u], mb[u]))
        return b
    objectives.Evaluator._whole = both
    data = synthetic.make_citeulike_like(content_len=200, vocab=8001)
    ep.run_case(data, model="basic_embedding", scheme=scheme, loss=loss, optimizer="sgd", epochs=1)


if __name__ == "__main__":
    main()
