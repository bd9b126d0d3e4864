"""CIDEr (n = 4, sigma = 6, idf from the evaluated references, x10) -- the acceptance metric of the path
(BASELINE.json: CIDEr within +-0.2).  Same interface as reference wikiweb2m/cider/cider.py:22-49
(`Cider().compute_score(gts, res) -> (corpus score, per-item scores)`); the arithmetic of cider_scorer.py:100-199 is
restated on sparse Counter vectors.  CPU, fp64."""
import math
from collections import Counter
from typing import Dict, List, Tuple

import numpy as np


def _ngrams(sentence: str, n: int = 4) -> Counter:
    words = sentence.split()
    c = Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            c[tuple(words[i:i + k])] += 1
    return c


class Cider:
    def __init__(self, test=None, refs=None, n: int = 4, sigma: float = 6.0):
        self._n = n
        self._sigma = sigma

    def method(self):
        return "CIDEr"

    def compute_score(self, gts: Dict, res: Dict) -> Tuple[float, np.ndarray]:
        assert gts.keys() == res.keys()
        n, sigma = self._n, self._sigma
        hyps, refs = [], []
        for key in gts.keys():
            hypo, ref = res[key], gts[key]
            assert type(hypo) is list and len(hypo) == 1
            assert type(ref) is list and len(ref) > 0
            hyps.append(_ngrams(hypo[0], n))
            refs.append([_ngrams(r, n) for r in ref])

        # document frequency: in how many items' reference sets an n-gram occurs (cider_scorer.py:100-111)
        df = Counter()
        for rs in refs:
            for ng in set(ng for r in rs for ng in r):
                df[ng] += 1
        assert len(hyps) >= max(df.values(), default=0)
        log_docs = np.log(float(len(refs)))

        def tfidf(counts: Counter):
            vec = [dict() for _ in range(n)]
            norm = [0.0] * n
            length = 0
            for ng, tf in counts.items():
                k = len(ng) - 1
                w = float(tf) * (log_docs - np.log(max(1.0, df.get(ng, 0.0))))
                vec[k][ng] = w
                norm[k] += w * w
                if k == 1:
                    length += tf            # the reference counts BIGRAMS as "length" (cider_scorer.py:137-138)
            return vec, [np.sqrt(x) for x in norm], length

        def similarity(vh, vr, nh, nr, lh, lr):
            delta = float(lh - lr)
            val = np.zeros(n)
            for k in range(n):
                for ng, w in vh[k].items():
                    wr = vr[k].get(ng, 0.0)
                    val[k] += min(w, wr) * wr                   # clipped
                if nh[k] != 0 and nr[k] != 0:
                    val[k] /= nh[k] * nr[k]
                assert not math.isnan(val[k])
                val[k] *= np.e ** (-(delta ** 2) / (2 * sigma ** 2))
            return val

        scores = []
        for h, rs in zip(hyps, refs):
            vh, nh, lh = tfidf(h)
            total = np.zeros(n)
            for r in rs:
                vr, nr, lr = tfidf(r)
                total += similarity(vh, vr, nh, nr, lh, lr)
            scores.append(np.mean(total) / len(rs) * 10.0)
        scores = np.array(scores)
        return float(np.mean(scores)), scores
