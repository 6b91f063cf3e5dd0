"""Generates tests/golden/known_answers.json: hand-derivable fp32 known-answer cases for the
aggregation stages.  All inputs are small dyadic rationals so every product and partial sum is exact
in fp32 and the expected outputs can be checked by hand (and are independent of summation order).
The expected values are computed here with exact rational arithmetic (fractions.Fraction) straight
from the formulas of the reference kernels -- NOT with the oracle or the HIP path:
  GCN  : Y[r,c] = sum_e val[e] * X[idx[e],c]                      (aggr_gcn.h:13-35)
  mean : Y / deg ; max : max_e val[e]*X[idx[e],c], 0 if empty     
  edgelist: (idx[e], row)                                         (aggregator.h:19-22)
  u_add_v : att[r,0] + att[idx[e],1]                              (aggr_gat.h:44-46)
Run:  python tests/golden/make_known_answers.py
"""
import json
import os
from fractions import Fraction as Fr

ptr = [0, 3, 3, 8, 9]
idx = [1, 2, 3, 0, 1, 2, 3, 0, 2]
val = [Fr(1, 2), Fr(-2), Fr(1, 4), Fr(3), Fr(1), Fr(-1, 2), Fr(2), Fr(1, 8), Fr(-4)]
X = [[Fr(1), Fr(-2), Fr(1, 2)], [Fr(3), Fr(1, 4), Fr(-1)], [Fr(-1, 2), Fr(2), Fr(4)], [Fr(8), Fr(-1, 8), Fr(1)]]
att = [[Fr(1, 2), Fr(-1)], [Fr(2), Fr(1, 4)], [Fr(-3), Fr(1)], [Fr(1, 8), Fr(-2)]]
V, F = 4, 3

gcn, mean, mx, edgelist, uaddv = [], [], [], [], []
for r in range(V):
    es = range(ptr[r], ptr[r + 1])
    row = [sum((val[e] * X[idx[e]][c] for e in es), Fr(0)) for c in range(F)]
    gcn.append(row)
    d = ptr[r + 1] - ptr[r]
    mean.append([v / d if d else Fr(0) for v in row])
    mx.append([max((val[e] * X[idx[e]][c] for e in es), default=Fr(0)) for c in range(F)])
    for e in es:
        edgelist += [idx[e], r]
        uaddv.append(att[r][0] + att[idx[e]][1])

f = lambda m: [[float(v) for v in row] for row in m]
out = {
    "_provenance": __doc__,
    "ptr": ptr, "idx": idx, "val": [float(v) for v in val], "X": f(X), "att": f(att),
    "gcn_sum": f(gcn), "gcn_sum_unit_weights": f([[sum((X[idx[e]][c] for e in range(ptr[r], ptr[r + 1])), Fr(0))
                                                    for c in range(F)] for r in range(V)]),
    "gcn_mean": f(mean), "gcn_max": f(mx), "edgelist": edgelist, "u_add_v": [float(v) for v in uaddv],
    "degrees": [ptr[r + 1] - ptr[r] for r in range(V)],
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "known_answers.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote known_answers.json")
