#!/usr/bin/env python3
"""Per-arm averages of the aggregation kernel from the rocprofv3 kernel trace of `bench.py --steps K --warmup W`: the bench
launches the same kernel for several inputs one after the other (no reorder, locality reorder = the headline arm, MinHash
clusters, uniform-random ids), each W warm-up + K timed launches + K launches with one event pair each (the median); the
probe launches are a different template instantiation.  usage: bench_arms_from_trace.py <kernel_trace.csv> K W"""
import csv
import sys

f, K, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = sorted((r for r in csv.DictReader(open(f)) if "k_gcn_plan" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
is_probe = lambda r: "false, true>" in r["Kernel_Name"] or "false, true, " in r["Kernel_Name"]   # k_gcn_plan<VEC, GROUP, IS_MAX, PROBE[, UNROLL]>
real = [r for r in rows if not is_probe(r)]
probe = [r for r in rows if is_probe(r)]


def arms(rs, names):
    out = []
    for i, name in enumerate(names):
        g = rs[i * (2 * K + W) + W:i * (2 * K + W) + W + K]   # per arm: W warm-up, K timed, K more for the per-launch median
        if len(g) == K:
            us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in g]
            out.append("%-52s n=%d avg %.2f us  median %.2f us  (%s)" % (name, K, sum(us) / K, sorted(us)[K // 2], g[0]["Kernel_Name"][:60]))
    return out


print("# rocprofv3 --kernel-trace of `python3 bench.py --steps %d --warmup %d --no-cpu`: timed launches per arm, in launch order" % (K, W))
for line in arms(real, ["no reorder", "locality reorder applied on load (HEADLINE arm)", "MinHash clusters, first-member order",
                        "uniform-random neighbor ids (same degrees)"]):
    print(line)
for line in arms(probe, ["gather probe of the headline arm (the measured ceiling)", "gather probe, uniform-random ids"]):
    print(line)
