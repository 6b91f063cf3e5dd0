// reorder.cpp -- locality reorder generator: the clustering of the reference's offline script
// (script/cluster2.py:29-171), restated as host C++ behind the C-ABI (gnnagg_cluster_reorder).
//
//   1. MinHash signature (num_perm hash functions) of every row's neighbor set          (cluster2.py:29-38)
//   2. LSH banding tuned for a Jaccard threshold (b bands x r rows minimising FP+FN area,
//      the rule datasketch's MinHashLSH uses); rows sharing a band bucket are candidates  (:79-94)
//   3. max-heap of candidate pairs keyed by the EXACT Jaccard similarity of the two rows'
//      neighbor lists (:45-50, :91)
//   4. greedy merge with path-halving union-find: a popped pair whose ends are both cluster roots is
//      merged (smaller into larger, ties into the first); a cluster reaching `cap` nodes is frozen;
//      a pair with a non-root end is re-queued as (root, root) with the roots' own similarity (:121-151)
//   5. nodes are written cluster by cluster, ascending inside a cluster (:156-171): rows[i] = old id placed at new
//      position i.  Cluster order: order_mode 0 = by each cluster's first node id (what the reference script does --
//      consecutive clusters are then unrelated); order_mode 1 = CACHE-AWARE GREEDY: the clusters are emitted so that each
//      next cluster is the one whose rows find the largest share of their source rows among the `cache_rows` most recently
//      gathered ones (an LRU model of an XCD's L2, which is what the reorder is for -- the reference's Figure 9 is an L2
//      hit-rate plot).  Clustering finds rows with SIMILAR neighbor sets; the ordering pass also puts clusters with
//      OVERLAPPING neighbor sets next to each other, which on the arxiv-shaped input is where most of the reuse is.
//
// Parity note: the reference hashes with datasketch (sha1 + its own permutations, README.md:16), whose
// source is not part of the reference tree, so the candidate sets -- and therefore the exact permutation --
// are not reproducible here ("parity unpinned" for reorder GENERATION).  Tests pin the
// algorithmic contract instead: valid permutation, planted clusters recovered, cluster cap respected.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"

namespace gnnagg {

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// (b, r) with b*r <= num_perm minimising false-positive + false-negative probability mass around `t`
static void lsh_params(double t, int num_perm, int *b_out, int *r_out)
{
    double best = 1e300;
    for (int b = 1; b <= num_perm; ++b)
        for (int r = 1; r <= num_perm / b; ++r) {
            double fp = 0, fn = 0;
            const double ds = 0.001;
            for (double s = ds / 2; s < 1.0; s += ds) {
                const double p = 1.0 - std::pow(1.0 - std::pow(s, r), b);
                if (s < t) fp += p * ds; else fn += (1.0 - p) * ds;
            }
            if (0.5 * fp + 0.5 * fn < best) { best = 0.5 * fp + 0.5 * fn; *b_out = b; *r_out = r; }
        }
}

static double jaccard_sorted(const int *a, int na, const int *b, int nb)
{
    if (na == 0 || nb == 0) return 0.0;
    int i = 0, j = 0, inter = 0, uni = 0;
    while (i < na && j < nb) {
        if (a[i] == b[j]) { ++inter; ++uni; const int v = a[i]; while (i < na && a[i] == v) ++i; while (j < nb && b[j] == v) ++j; }
        else if (a[i] < b[j]) { ++uni; const int v = a[i]; while (i < na && a[i] == v) ++i; }
        else { ++uni; const int v = b[j]; while (j < nb && b[j] == v) ++j; }
    }
    while (i < na) { ++uni; const int v = a[i]; while (i < na && a[i] == v) ++i; }
    while (j < nb) { ++uni; const int v = b[j]; while (j < nb && b[j] == v) ++j; }
    return (double)inter / (double)uni;
}

// Emits the clusters (members[c] = node ids, ascending) in cache-aware greedy order.
// Rows longer than kLongRow edges take no part in the cache model: the balanced kernels cut them into segments that run
// first and on whatever XCD the dispatcher picks, not in row order beside their neighbors (27 % of the edges of the
// arxiv-shaped input); clusters made only of such rows are written last.
static constexpr int kLongRow = 64;
// Source rows gathered by more than kHubCut clusters do not vote: they are in the cache whatever the order is (or thrash
// it whatever the order is), and scanning their long gatherer lists on every (re)entry is what makes the pass quadratic
// on power-law graphs (products-shaped: hours -> minutes).  They still occupy cache capacity in the model.
static const long kHubCut = 4096;

static void emit_cache_greedy(const int *ptr, const int *idx, int V, const std::vector<std::vector<int>> &members,
                              int cache_rows, int *rows_out)
{
    const int NC = (int)members.size();
    const long E = ptr[V];
    // cluster -> (source, multiplicity) lists, and their transpose source -> (cluster, multiplicity)
    std::vector<long> cp((size_t)NC + 1, 0);
    std::vector<int> cs, cm;
    cs.reserve((size_t)E);
    cm.reserve((size_t)E);
    {
        std::vector<int> tmp;
        for (int c = 0; c < NC; ++c) {
            tmp.clear();
            for (int v : members[c])
                if (ptr[v + 1] - ptr[v] <= kLongRow) tmp.insert(tmp.end(), idx + ptr[v], idx + ptr[v + 1]);
            std::sort(tmp.begin(), tmp.end());
            for (size_t a = 0; a < tmp.size();) {
                size_t b = a;
                while (b < tmp.size() && tmp[b] == tmp[a]) ++b;
                cs.push_back(tmp[a]);
                cm.push_back((int)(b - a));
                a = b;
            }
            cp[c + 1] = (long)cs.size();
        }
    }
    const long NZ = (long)cs.size();
    std::vector<long> tp((size_t)V + 1, 0);
    for (long k = 0; k < NZ; ++k) ++tp[(size_t)cs[k] + 1];
    for (int v = 0; v < V; ++v) tp[v + 1] += tp[v];
    std::vector<int> tc((size_t)NZ), tm((size_t)NZ);
    {
        std::vector<long> cur(tp.begin(), tp.end() - 1);
        for (int c = 0; c < NC; ++c)
            for (long k = cp[c]; k < cp[c + 1]; ++k) {
                const long pos = cur[cs[k]]++;
                tc[pos] = c;
                tm[pos] = cm[k];
            }
    }
    std::vector<long> edges((size_t)NC);
    for (int c = 0; c < NC; ++c) {
        long e = 0;
        for (int v : members[c])
            if (ptr[v + 1] - ptr[v] <= kLongRow) e += ptr[v + 1] - ptr[v];
        edges[c] = std::max(1L, e);
    }
    // LRU over source rows: intrusive list, head = most recent
    std::vector<int> prv((size_t)V, -1), nxt((size_t)V, -1);
    std::vector<char> cached((size_t)V, 0), placed((size_t)NC, 0);
    int head = -1, tail = -1, ncached = 0;
    std::vector<long> score((size_t)NC, 0);
    // Candidates in a bucket queue keyed by the cached share score / edges (kBuckets levels; O(1) per score change, O(NC)
    // memory: a binary heap with lazy deletion grew to tens of GB on the products-shaped graph).  Inside a bucket the most
    // recently touched cluster goes first.
    constexpr int kBuckets = 128;
    std::vector<int> bprev((size_t)NC, -1), bnext((size_t)NC, -1), inb((size_t)NC, -1), bhead(kBuckets + 1, -1);
    int top = -1;
    auto bucket_of = [&](int c) -> int {
        if (score[c] <= 0) return -1;
        const long b = score[c] * kBuckets / edges[c];
        return (int)std::min<long>(b, kBuckets);
    };
    auto b_unlink = [&](int c) {
        const int b = inb[c];
        if (b < 0) return;
        if (bprev[c] >= 0) bnext[bprev[c]] = bnext[c]; else bhead[b] = bnext[c];
        if (bnext[c] >= 0) bprev[bnext[c]] = bprev[c];
        inb[c] = -1;
    };
    auto b_update = [&](int c) {  // after a score change: move c to the front of its (new) bucket
        const int b = bucket_of(c);
        b_unlink(c);
        if (b < 0) return;
        bprev[c] = -1; bnext[c] = bhead[b];
        if (bhead[b] >= 0) bprev[bhead[b]] = c;
        bhead[b] = c;
        inb[c] = b;
        if (b > top) top = b;
    };
    auto unlink = [&](int s) {
        if (prv[s] >= 0) nxt[prv[s]] = nxt[s]; else head = nxt[s];
        if (nxt[s] >= 0) prv[nxt[s]] = prv[s]; else tail = prv[s];
    };
    auto to_front = [&](int s) {
        prv[s] = -1; nxt[s] = head;
        if (head >= 0) prv[head] = s;
        head = s;
        if (tail < 0) tail = s;
    };
    auto touch = [&](int s) {
        if (cached[s]) { unlink(s); to_front(s); return; }
        cached[s] = 1; ++ncached;
        to_front(s);
        if (tp[s + 1] - tp[s] <= kHubCut)
            for (long k = tp[s]; k < tp[s + 1]; ++k)
                if (!placed[tc[k]]) { score[tc[k]] += tm[k]; b_update(tc[k]); }
        if (ncached > cache_rows) {
            const int old = tail;
            unlink(old);
            cached[old] = 0; --ncached;
            if (tp[old + 1] - tp[old] <= kHubCut)
                for (long k = tp[old]; k < tp[old + 1]; ++k)
                    if (!placed[tc[k]]) { score[tc[k]] -= tm[k]; b_update(tc[k]); }
        }
    };
    int pos = 0, seed = 0, seed2 = 0;
    for (int done = 0; done < NC; ++done) {
        int c = -1;
        while (top >= 0 && bhead[top] < 0) --top;
        if (top >= 0) c = bhead[top];
        if (c < 0) {  // nothing related to the cache: next unplaced cluster in first-member order (source-less ones last)
            while (seed < NC && (placed[seed] || cp[seed] == cp[seed + 1])) ++seed;
            if (seed < NC) {
                c = seed;
            } else {
                while (placed[seed2]) ++seed2;
                c = seed2;
            }
        }
        b_unlink(c);
        placed[c] = 1;
        for (int v : members[c]) rows_out[pos++] = v;
        for (long k = cp[c]; k < cp[c + 1]; ++k) touch(cs[k]);
    }
}

// int -> T open-addressing hash map (linear probing, backward-shift deletion, power-of-two capacity): the walkers' candidate
// tables see billions of operations on graphs of the products-shaped size, where std::unordered_map's node allocations cost 4x.
template <class T>
struct IntMap {
    std::vector<int> keys;   // -1 = empty
    std::vector<T> vals;
    size_t mask = 0, n = 0;
    explicit IntMap(size_t cap_pow2 = 1 << 16) : keys(cap_pow2, -1), vals(cap_pow2), mask(cap_pow2 - 1) {}
    static size_t hash(int k) { return (size_t)((uint32_t)k * 2654435761u); }
    T *find(int k)
    {
        for (size_t i = hash(k) & mask;; i = (i + 1) & mask) {
            if (keys[i] == k) return &vals[i];
            if (keys[i] < 0) return nullptr;
        }
    }
    T *insert(int k, const T &v)   // k must not be present
    {
        if ((n + 1) * 2 > keys.size()) grow();
        size_t i = hash(k) & mask;
        while (keys[i] >= 0) i = (i + 1) & mask;
        keys[i] = k; vals[i] = v; ++n;
        return &vals[i];
    }
    void erase(int k)
    {
        size_t i = hash(k) & mask;
        while (keys[i] != k) { if (keys[i] < 0) return; i = (i + 1) & mask; }
        --n;
        for (size_t j = (i + 1) & mask;; j = (j + 1) & mask) {   // backward shift: keep every probe sequence unbroken
            if (keys[j] < 0) break;
            const size_t home = hash(keys[j]) & mask;
            if (((j - home) & mask) >= ((j - i) & mask)) { keys[i] = keys[j]; vals[i] = vals[j]; i = j; }
        }
        keys[i] = -1;
    }
    void grow()
    {
        std::vector<int> ok;
        std::vector<T> ov;
        ok.swap(keys); ov.swap(vals);
        keys.assign(ok.size() * 2, -1); vals.resize(ok.size() * 2); mask = keys.size() - 1; n = 0;
        for (size_t i = 0; i < ok.size(); ++i)
            if (ok[i] >= 0) insert(ok[i], ov[i]);
    }
};

// The same greedy with SEVERAL WALKERS for graphs where the serial pass takes minutes (products-shaped: 5.4 minutes of one core).
// Every walker runs the serial algorithm with its OWN state -- its own LRU model of an L2, its own scores and bucket queue
// (hash-mapped: only the clusters its cached sources vote for have an entry) -- and the walkers share nothing but the read-only
// lists and one `placed` flag per cluster.  A walker without candidates takes the next unplaced cluster of its stripe of the
// cluster list as a new seed.  The walkers' sequences are concatenated; source-less clusters follow.  (A first version let a
// cluster be the candidate of ONE walker at a time so that the dense score arrays could be shared: a popular source entering one
// walker's cache then claimed thousands of clusters all over the graph and hid them from the walker they belonged to -- window
// footprint 0.44 serial, 0.49 / 0.54 / 0.63 with 2 / 4 / 8 walkers.  Round 3 ran one free-running walker per thread with an
// atomic exchange on `placed`: the order depended on thread timing and on the thread count; the rounds below remove both.)
static void emit_cache_greedy_parallel(const int *ptr, const int *idx, int V, const std::vector<std::vector<int>> &members,
                                       int cache_rows, int walkers, int *rows_out)
{
    const int NC = (int)members.size();
    const long E = ptr[V];
    // cluster -> (source, multiplicity) and its transpose, as in emit_cache_greedy (built in parallel here)
    std::vector<long> cp((size_t)NC + 1, 0);
    std::vector<long> edges((size_t)NC);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int c = 0; c < NC; ++c) {
        std::vector<int> tmp;
        long e = 0;
        for (int v : members[c])
            if (ptr[v + 1] - ptr[v] <= kLongRow) { tmp.insert(tmp.end(), idx + ptr[v], idx + ptr[v + 1]); e += ptr[v + 1] - ptr[v]; }
        std::sort(tmp.begin(), tmp.end());
        long d = 0;
        for (size_t a = 0; a < tmp.size();) { size_t b = a; while (b < tmp.size() && tmp[b] == tmp[a]) ++b; ++d; a = b; }
        cp[c + 1] = d;
        edges[c] = std::max(1L, e);
    }
    for (int c = 0; c < NC; ++c) cp[c + 1] += cp[c];
    const long NZ = cp[NC];
    std::vector<int> cs((size_t)NZ), cm((size_t)NZ);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int c = 0; c < NC; ++c) {
        std::vector<int> tmp;
        for (int v : members[c])
            if (ptr[v + 1] - ptr[v] <= kLongRow) tmp.insert(tmp.end(), idx + ptr[v], idx + ptr[v + 1]);
        std::sort(tmp.begin(), tmp.end());
        long k = cp[c];
        for (size_t a = 0; a < tmp.size();) {
            size_t b = a;
            while (b < tmp.size() && tmp[b] == tmp[a]) ++b;
            cs[k] = tmp[a]; cm[k] = (int)(b - a); ++k;
            a = b;
        }
    }
    (void)E;
    std::vector<long> tp((size_t)V + 1, 0);
    for (long k = 0; k < NZ; ++k) ++tp[(size_t)cs[k] + 1];
    for (int v = 0; v < V; ++v) tp[v + 1] += tp[v];
    std::vector<int> tc((size_t)NZ), tm((size_t)NZ);
    {
        std::vector<long> cur(tp.begin(), tp.end() - 1);
        for (int c = 0; c < NC; ++c)
            for (long k = cp[c]; k < cp[c + 1]; ++k) {
                const long pos = cur[cs[k]]++;
                tc[pos] = c;
                tm[pos] = cm[k];
            }
    }
    constexpr int kBuckets = 128;
    // Bulk-synchronous rounds make the result a function of the input and the walker count alone: inside a round every walker
    // sees the `placed` flags as they stood when the round began plus its own claims (a bit per cluster), and places up to
    // kRoundSteps clusters; between rounds ONE thread merges the claims in walker order -- a cluster two walkers claimed in the same
    // round stays with the lower walker and leaves the other's sequence (its sources stay in that walker's cache model: harmless
    // and, like everything else here, independent of timing).  New seeds come from the walker's own stripe of the cluster list
    // first, then from the stripes after it.  The walkers are logical: a round is an `omp for` over them, so the thread count
    // changes the speed and not the order.
#ifndef GNNAGG_REORDER_ROUND
#define GNNAGG_REORDER_ROUND 256
#endif
    constexpr int kRoundSteps = GNNAGG_REORDER_ROUND;
    std::vector<char> placed((size_t)NC, 0);   // written between rounds only
    struct Cand { long score; int bprev, bnext, inb; };
    struct Walker {
        IntMap<Cand> cand{(size_t)1 << 14};    // clusters this walker's cached sources vote for
        std::vector<int> bhead = std::vector<int>(kBuckets + 1, -1);
        int top = -1;
        // LRU over source rows of THIS walker: slots in a ring of cache_rows + 1 entries, key -> slot in a hash map
        std::unordered_map<int, int> slot_of;
        std::vector<int> skey, sprev, snext, free_slots;
        int head = -1, tail = -1;
        std::vector<uint64_t> mine;            // claims of the current round
        std::vector<int> claims, seq, stripe_cur;
    };
    std::vector<Walker> ws((size_t)walkers);
    std::vector<int> stripe_lo((size_t)walkers + 1);
    for (int w = 0; w <= walkers; ++w) stripe_lo[w] = (int)((long)NC * w / walkers);
#pragma omp parallel for schedule(static, 1)
    for (int w = 0; w < walkers; ++w) {
        Walker &W = ws[w];
        W.slot_of.reserve((size_t)cache_rows * 2);
        W.skey.resize((size_t)cache_rows + 1);
        W.sprev.assign((size_t)cache_rows + 1, -1);
        W.snext.assign((size_t)cache_rows + 1, -1);
        for (int i = cache_rows; i >= 0; --i) W.free_slots.push_back(i);
        W.mine.assign(((size_t)NC + 63) / 64, 0);
        W.stripe_cur.resize((size_t)walkers);
        for (int j = 0; j < walkers; ++j) W.stripe_cur[j] = stripe_lo[j];
    }
    auto run_round = [&](int w) {
        Walker &W = ws[w];
        IntMap<Cand> &cand = W.cand;
        std::vector<int> &bhead = W.bhead;
        auto taken = [&](int c) { return placed[c] || ((W.mine[(size_t)c >> 6] >> (c & 63)) & 1); };
        auto b_unlink = [&](Cand &e) {
            const int b = e.inb;
            if (b < 0) return;
            if (e.bprev >= 0) cand.find(e.bprev)->bnext = e.bnext; else bhead[b] = e.bnext;
            if (e.bnext >= 0) cand.find(e.bnext)->bprev = e.bprev;
            e.inb = -1;
        };
        auto b_link = [&](int c, Cand &e) {   // to the front of the bucket its score belongs to
            const int b = e.score <= 0 ? -1 : (int)std::min<long>(e.score * kBuckets / edges[c], kBuckets);
            if (b < 0) return;
            e.bprev = -1; e.bnext = bhead[b];
            if (bhead[b] >= 0) cand.find(bhead[b])->bprev = c;
            bhead[b] = c;
            e.inb = b;
            if (b > W.top) W.top = b;
        };
        auto vote = [&](int s, int sign) {   // source s entered (+1) / left (-1) this walker's cache
            if (tp[s + 1] - tp[s] > kHubCut) return;
            for (long k = tp[s]; k < tp[s + 1]; ++k) {
                const int c = tc[k];
                if (taken(c)) continue;
                Cand *e = cand.find(c);
                if (!e) {
                    if (sign < 0) continue;
                    e = cand.insert(c, Cand{0, -1, -1, -1});
                }
                b_unlink(*e);
                e->score += sign * (long)tm[k];
                if (e->score <= 0) cand.erase(c);
                else b_link(c, *e);
            }
        };
        auto lru_unlink = [&](int sl) {
            if (W.sprev[sl] >= 0) W.snext[W.sprev[sl]] = W.snext[sl]; else W.head = W.snext[sl];
            if (W.snext[sl] >= 0) W.sprev[W.snext[sl]] = W.sprev[sl]; else W.tail = W.sprev[sl];
        };
        auto lru_front = [&](int sl) {
            W.sprev[sl] = -1; W.snext[sl] = W.head;
            if (W.head >= 0) W.sprev[W.head] = sl;
            W.head = sl;
            if (W.tail < 0) W.tail = sl;
        };
        auto touch = [&](int s) {
            auto it = W.slot_of.find(s);
            if (it != W.slot_of.end()) { lru_unlink(it->second); lru_front(it->second); return; }
            const int sl = W.free_slots.back();
            W.free_slots.pop_back();
            W.skey[sl] = s;
            W.slot_of.emplace(s, sl);
            lru_front(sl);
            vote(s, +1);
            if ((int)W.slot_of.size() > cache_rows) {
                const int old = W.tail;
                lru_unlink(old);
                const int so = W.skey[old];
                W.slot_of.erase(so);
                W.free_slots.push_back(old);
                vote(so, -1);
            }
        };
        for (int step = 0; step < kRoundSteps; ++step) {
            int c = -1;
            for (;;) {   // best candidate nobody had placed when the round began
                while (W.top >= 0 && bhead[W.top] < 0) --W.top;
                if (W.top < 0) break;
                const int t = bhead[W.top];
                b_unlink(*cand.find(t));
                cand.erase(t);
                if (!taken(t)) { c = t; break; }
            }
            if (c < 0) {   // no candidate related to this walker's cache: a new seed, own stripe first
                for (int j = 0; j < walkers && c < 0; ++j) {
                    const int st = (w + j) % walkers;
                    int &cur = W.stripe_cur[st];
                    while (cur < stripe_lo[st + 1] && (cp[cur] == cp[cur + 1] || taken(cur))) ++cur;   // source-less clusters are written last
                    if (cur < stripe_lo[st + 1]) c = cur++;
                }
                if (c < 0) break;
                if (Cand *e = cand.find(c)) { b_unlink(*e); cand.erase(c); }
            }
            W.mine[(size_t)c >> 6] |= 1ULL << (c & 63);
            W.claims.push_back(c);
            for (long k = cp[c]; k < cp[c + 1]; ++k) touch(cs[k]);
        }
    };
    for (;;) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int w = 0; w < walkers; ++w) run_round(w);
        long claimed = 0;
        for (int w = 0; w < walkers; ++w) {   // merge in walker order
            Walker &W = ws[w];
            claimed += (long)W.claims.size();
            for (int c : W.claims) {
                W.mine[(size_t)c >> 6] = 0;
                if (!placed[c]) { placed[c] = 1; W.seq.push_back(c); }
            }
            W.claims.clear();
        }
        if (claimed == 0) break;
    }
    int pos = 0;
    for (int w = 0; w < walkers; ++w)
        for (int c : ws[w].seq)
            for (int v : members[c]) rows_out[pos++] = v;
    long left = 0;
    for (int c = 0; c < NC; ++c)
        if (!placed[c]) {   // the source-less clusters, in first-member order
            ++left;
            for (int v : members[c]) rows_out[pos++] = v;
        }
}

int cluster_reorder(const int *ptr, const int *idx, int V, double threshold, int num_perm, int cap, uint64_t seed,
                    int max_bucket, int *rows_out, int *num_clusters_out, int order_mode, int cache_rows)
{
    if (V < 0 || num_perm < 1 || num_perm > 256 || cap < 1) return fail(GNNAGG_ERR_ARG, "bad cluster_reorder arguments");
    if (order_mode < 0 || order_mode > 1 || (order_mode == 1 && cache_rows < 1)) return fail(GNNAGG_ERR_ARG, "bad cluster order mode");
    // sorted neighbor sets (the similarity is on sets; the CSR itself is not modified)
    std::vector<int> sidx(idx, idx + ptr[V]);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < V; ++i) std::sort(sidx.begin() + ptr[i], sidx.begin() + ptr[i + 1]);

    int B = 1, R = 1;
    lsh_params(threshold, num_perm, &B, &R);
    std::vector<uint64_t> ha(num_perm), hb(num_perm);
    for (int k = 0; k < num_perm; ++k) { ha[k] = mix64(seed + 2 * k + 1) | 1ULL; hb[k] = mix64(seed + 2 * k + 2); }

    // band keys per row
    std::vector<uint64_t> bandkey((size_t)V * B);
#pragma omp parallel
    {
        std::vector<uint64_t> sig(num_perm);
#pragma omp for schedule(dynamic, 1024)
        for (int i = 0; i < V; ++i) {
            std::fill(sig.begin(), sig.end(), ~0ULL);
            for (int e = ptr[i]; e < ptr[i + 1]; ++e) {
                const uint64_t x = mix64((uint64_t)(uint32_t)sidx[e] + 0x9e3779b97f4a7c15ULL);
                for (int k = 0; k < num_perm; ++k) {
                    const uint64_t h = mix64(x * ha[k] + hb[k]);
                    if (h < sig[k]) sig[k] = h;
                }
            }
            for (int b = 0; b < B; ++b) {
                uint64_t key = 0x1234567ULL + b;
                for (int r = 0; r < R; ++r) key = mix64(key ^ sig[b * R + r]);
                bandkey[(size_t)i * B + b] = key;
            }
        }
    }

    struct Pair { double simi; int p1, p2; };
    const bool no_merge = cap <= 1;  // cap 1: every node is its own cluster (the ordering pass alone)
    auto cmp = [](const Pair &a, const Pair &b) { return a.simi < b.simi; };
    std::priority_queue<Pair, std::vector<Pair>, decltype(cmp)> que(cmp);
    std::unordered_set<uint64_t> queued;
    auto makenum = [&](int a, int b) { if (a > b) std::swap(a, b); return (uint64_t)a * (uint64_t)V + (uint64_t)b; };
    auto simi = [&](int a, int b) {
        return jaccard_sorted(sidx.data() + ptr[a], ptr[a + 1] - ptr[a], sidx.data() + ptr[b], ptr[b + 1] - ptr[b]);
    };
    // candidates: rows sharing a band bucket (empty rows are never queried, cluster2.py:83-84)
    if (!no_merge) {
        std::vector<std::pair<uint64_t, int>> keys;
        keys.reserve(V);
        for (int b = 0; b < B; ++b) {
            keys.clear();
            for (int i = 0; i < V; ++i)
                if (ptr[i] != ptr[i + 1]) keys.emplace_back(bandkey[(size_t)i * B + b], i);
            std::sort(keys.begin(), keys.end());
            for (size_t s = 0; s < keys.size();) {
                size_t t = s;
                while (t < keys.size() && keys[t].first == keys[s].first) ++t;
                // Small buckets contribute all their pairs (what an LSH query returns).  A bucket shared by very many
                // rows (dense graphs: reddit-shaped rows all overlap a little) would be quadratic in time and memory,
                // so there every row is only paired with the next `max_bucket` rows of the bucket; rows still meet
                // through the other bands and through the root-to-root re-queueing of the merge phase.
                const size_t nb = t - s;
                for (size_t u = s; u < t; ++u) {
                    const size_t lim = nb <= 16 ? t : std::min(t, u + 1 + (size_t)max_bucket);
                    for (size_t v = u + 1; v < lim; ++v) {
                        const uint64_t id = makenum(keys[u].second, keys[v].second);
                        if (queued.insert(id).second) que.push({simi(keys[u].second, keys[v].second), keys[u].second, keys[v].second});
                    }
                }
                s = t;
            }
        }
    }

    std::vector<int> cluster_id(V), cluster_sz(V, 1);
    std::vector<char> deleted(V, 0);
    for (int i = 0; i < V; ++i) cluster_id[i] = i;
    auto root = [&](int i) {
        while (i != cluster_id[i]) { cluster_id[i] = cluster_id[cluster_id[i]]; i = cluster_id[i]; }
        return i;
    };
    while (!que.empty()) {
        const Pair it = que.top();
        que.pop();
        int p1 = it.p1, p2 = it.p2;
        queued.erase(makenum(p1, p2));
        if (p1 == cluster_id[p1] && p2 == cluster_id[p2]) {
            if (deleted[p1] || deleted[p2]) continue;
            if (cluster_sz[p1] < cluster_sz[p2]) std::swap(p1, p2);  // merge the smaller (p2) into p1; ties keep p1
            cluster_id[p2] = p1;
            cluster_sz[p1] += cluster_sz[p2];
            if (cluster_sz[p1] >= cap) deleted[p1] = 1;
        } else {
            p1 = root(p1);
            p2 = root(p2);
            if (deleted[p1] || deleted[p2] || p1 == p2) continue;
            if (queued.insert(makenum(p1, p2)).second) que.push({simi(p1, p2), p1, p2});
        }
    }
    // clusters in order of first appearance, members ascending
    std::vector<int> first_pos(V, -1), order;
    std::vector<std::vector<int>> members;
    for (int i = 0; i < V; ++i) {
        const int ro = root(i);
        if (first_pos[ro] < 0) { first_pos[ro] = (int)members.size(); members.emplace_back(); }
        members[first_pos[ro]].push_back(i);
    }
    if (order_mode == 1) {
        // several walkers from the size on where the serial pass takes tens of seconds (GNNAGG_REORDER_WALKERS: 1 = always the
        // serial pass, N = N logical walkers; either way the order is deterministic)
        const int walkers_env = getenv("GNNAGG_REORDER_WALKERS") ? atoi(getenv("GNNAGG_REORDER_WALKERS")) : 0;  // (read per call: tests switch it)
        // (a fixed count, not the thread count: the order must not depend on the machine it was computed on)
        int walkers = walkers_env > 0 ? walkers_env : (ptr[V] >= 20000000 ? 64 : 1);
        walkers = std::max(1, std::min(walkers, (int)members.size() / 4096 + 1));
        if (walkers > 1) emit_cache_greedy_parallel(ptr, idx, V, members, cache_rows, walkers, rows_out);
        else emit_cache_greedy(ptr, idx, V, members, cache_rows, rows_out);
    } else {
        int pos = 0;
        for (auto &mset : members) for (int v : mset) rows_out[pos++] = v;
    }
    if (num_clusters_out) *num_clusters_out = (int)members.size();
    return GNNAGG_OK;
}

}  // namespace gnnagg
