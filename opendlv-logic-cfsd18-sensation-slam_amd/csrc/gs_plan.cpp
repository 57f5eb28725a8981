// gs_plan.cpp — structure phase: elimination order + symbolic multifrontal plan (host only).
//
// Replaces, for the HIP back-end, what the reference runs once per optimize() call inside g2o/Eigen:
// initializeOptimization + BlockSolver::buildStructure (index maps, block pattern; call site
// reference src/slam.cpp:480) and Eigen's analyzePattern (AMD ordering + elimination tree,
// reference thirdparty/Eigen/src/OrderingMethods/Amd.h:94,
// thirdparty/Eigen/src/SparseCholesky/SimplicialCholesky_impl.h:51-98).
//
// MI355X-first design instead of AMD + up-looking simplicial columns (a strictly sequential column
// loop): the cone-track graph is a chain of poses, each tied to the few cones in view, so a vertex
// separator of the joint pose+cone graph is tiny (one pose + the ~8 cones seen from both sides,
// ~19 scalars) while eliminating the cones first would leave a dense pose band hundreds wide.
// Nested dissection over the temporal pose order therefore yields a balanced assembly tree of small
// dense fronts (~30-60 scalars) with log2(N) levels: thousands of independent fronts per level for the
// 256 CUs, dense-tile arithmetic inside a front (LDS resident, fp64 MFMA-shaped), no atomics.
// The construction is a heuristic for ORDER only; the symbolic factorisation below is exact for any
// graph, so every ordering gives the same solution as the reference's joint Cholesky up to rounding.
#include "gs_host.hpp"
#include "gs_parallel.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace gs {

void HostGraph::clear() {
    pose_id.clear(); pose_est.clear(); pose_fixed.clear(); lm_id.clear(); lm_est.clear(); lm_fixed.clear();
    pose_index.clear(); lm_index.clear();
    pp_i.clear(); pp_j.clear(); pp_z.clear(); pp_info.clear();
    pl_p.clear(); pl_l.clear(); pl_z.clear(); pl_info.clear();
    ++structure_version; ++estimate_version; ++reshape_version;
}

namespace {

// fill on the host threads; a vector that already has the size (a handle that plans again: build_plan's recycled arrays) is not re-initialised first
template <class V, class T> void pfill(V &v, size_t n, T value) {
    if (v.size() != n) { v.clear(); v.reserve(n); hint_huge(v); v.resize(n); }
    auto *p = v.data();
    parallel_chunks((int64_t)n, 1 << 18, [&](int64_t b0, int64_t e0, int) { std::fill(p + b0, p + e0, value); });
}

// resize / assign of a big array with the huge-page hint placed before its first touch
template <class V> void big_resize(V &v, size_t n) { if (v.capacity() < n) { v.clear(); v.reserve(n); hint_huge(v); } v.resize(n); }
template <class V, class T> void big_assign(V &v, size_t n, T value) { if (v.capacity() < n) { v.clear(); v.reserve(n); hint_huge(v); } v.assign(n, value); }
// counts -> inclusive running sums in place, v[i] += v[i - 1] for i = 1 .. n (v has n + 1 entries), on the host threads: a pose-window shard of
// 8 x 100k poses has half a dozen of these over 0.8 M entries
inline void parallel_prefix(int32_t *v, int64_t n) {
    const int T = chunk_count(n, 16384);
    if (T <= 1) { for (int64_t i = 1; i <= n; ++i) v[i] += v[i - 1]; return; }
    std::vector<int64_t> part((size_t)T + 1, 0);
    parallel_chunks(n, 16384, [&](int64_t b0, int64_t e0, int t) { int64_t a = 0; for (int64_t i = b0 + 1; i <= e0; ++i) a += v[i]; part[(size_t)t + 1] = a; });
    part[0] = v[0];
    for (int t = 0; t < T; ++t) part[(size_t)t + 1] += part[(size_t)t];
    parallel_chunks(n, 16384, [&](int64_t b0, int64_t e0, int t) { int32_t run = (int32_t)part[(size_t)t]; for (int64_t i = b0 + 1; i <= e0; ++i) { run += v[i]; v[i] = run; } });
}

struct Builder {
    const HostGraph &g;
    PlanOptions opt;
    int nfp = 0, nfl = 0, nv = 0;              // free poses, free landmarks, free vertices
    std::vector<int32_t> fp_of_pose, fl_of_lm; // insertion index -> free index (-1 fixed)
    std::vector<int32_t> pose_of_fp, lm_of_fl;
    // vertex-level adjacency with edge references (both endpoints free)
    struct Inc { int32_t other; int32_t epos; int32_t kind; };   // kind: 0 pp (this is i), 1 pp (this is j), 2 pl (this is pose), 3 pl (this is lm)
    std::vector<int32_t> inc_start; std::unique_ptr<Inc[]> inc_store; size_t inc_cap = 0; Inc *inc = nullptr;   // (not a vector: 21 MB at 100k poses that need no zero fill)
    // ND helpers
    std::vector<int32_t> cone_obs_start, cone_obs;   // per free landmark: sorted free-pose positions
    std::vector<int32_t> obs_lo, obs_hi;             // ... and the first / last of them
    std::vector<uint8_t> assigned;
    std::vector<uint8_t> opaque_pose;                // pose-window shards: the pose belongs to an opaque supernode (a whole subtree of another rank)
    int window(int fpos) const { return (int)((int64_t)fpos * std::max(1, opt.world) / std::max(1, nfp)); }

    explicit Builder(const HostGraph &gg, const PlanOptions &o) : g(gg), opt(o) {}
    // the buffers of the previous plan build of this handle (a workspace: no allocation, no first-touch page faults, nothing to free on the way out)
    void adopt(Builder &o) {
#define GS_ADOPT(m) do { m = std::move(o.m); m.clear(); } while (0)
        fp_of_pose = std::move(o.fp_of_pose); fl_of_lm = std::move(o.fl_of_lm);      // (rewritten in full by index_vertices: the size stays)
        pose_of_fp = std::move(o.pose_of_fp); lm_of_fl = std::move(o.lm_of_fl); GS_ADOPT(inc_start); GS_ADOPT(cone_obs_start); GS_ADOPT(cone_obs);
        GS_ADOPT(obs_lo); GS_ADOPT(obs_hi); GS_ADOPT(assigned); GS_ADOPT(opaque_pose); GS_ADOPT(sn); GS_ADOPT(lazy_pose); GS_ADOPT(lazy_lm);
        GS_ADOPT(wf); GS_ADOPT(seen_nb); GS_ADOPT(seen_b); GS_ADOPT(opaque_of_pose); GS_ADOPT(pp_touch);
#undef GS_ADOPT
        inc_store = std::move(o.inc_store); inc_cap = o.inc_cap; o.inc_cap = 0;
    }

    int dim(int v) const { return v < nfp ? 3 : 2; }

    void index_vertices() {                                            // (on the host threads: counts per part, then every part numbers its own)
        auto index = [](const std::vector<uint8_t> &fixed, std::vector<int32_t> &free_of, std::vector<int32_t> &of_free) {
            const int64_t n = (int64_t)fixed.size();
            if (free_of.size() != (size_t)n) { free_of.clear(); free_of.resize((size_t)n); }
            std::vector<int32_t> cnt((size_t)chunk_count(n, 16384) + 1, 0);
            parallel_chunks(n, 16384, [&](int64_t b0, int64_t e0, int t) { int32_t c = 0; for (int64_t i = b0; i < e0; ++i) c += !fixed[(size_t)i]; cnt[(size_t)t + 1] = c; });
            for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
            if (of_free.size() != (size_t)cnt.back()) { of_free.clear(); of_free.resize((size_t)cnt.back()); }
            parallel_chunks(n, 16384, [&](int64_t b0, int64_t e0, int t) { int32_t c = cnt[(size_t)t];
                for (int64_t i = b0; i < e0; ++i) { if (fixed[(size_t)i]) free_of[(size_t)i] = -1; else { free_of[(size_t)i] = c; of_free[(size_t)c++] = (int32_t)i; } } });
            return (int)cnt.back(); };
        nfp = index(g.pose_fixed, fp_of_pose, pose_of_fp); nfl = index(g.lm_fixed, fl_of_lm, lm_of_fl);
        nv = nfp + nfl;
    }

    // Per-vertex incidence lists straight from the CSR arrays of the layout phase (pose -> its odometry incidences
    // and observation edges, landmark -> its edges): every vertex fills its own range, so the build runs on all host
    // threads.  Order inside a vertex: odometry edges (insertion order), then observation edges (insertion order for a
    // pose, pose order for a landmark).
    // Pose-window shards: the incidence list of a pose of ANOTHER rank's window is not stored (7/8 of the pose side at world 8); the few
    // places that need one — the top-level splits, the boundary of an opaque supernode, a shared separator's records — enumerate it from the
    // grouped edge arrays, in the same order.
    // The same for a landmark all of whose observers lie in one window of another rank (it lives inside that rank's opaque supernode, or in a
    // separator a split inside that window makes): its observer positions are kept (the dissection asks for them), its list is not.
    const Plan *plan_ = nullptr; const std::vector<int32_t> *lm_k_ = nullptr;
    std::vector<uint8_t> lazy_pose, lazy_lm;
    template <class F> void for_inc(int v, F &&fn) const {
        if (v >= nfp ? !lazy_lm[v - nfp] : !lazy_pose[v]) { for (int q = inc_start[v]; q < inc_start[v + 1]; ++q) fn(inc[q]); return; }
        const Plan &P = *plan_;
        if (v >= nfp) { const int l = lm_of_fl[v - nfp];
            for (int q = P.lm_start[l]; q < P.lm_start[l + 1]; ++q) { const int k = (*lm_k_)[q]; const int fp = fp_of_pose[g.pl_p[k]];
                if (fp >= 0) fn(Inc{fp, k, 3}); }
            return; }
        const int p = pose_of_fp[v];
        for (int q = P.ppadj_start[p]; q < P.ppadj_start[p + 1]; ++q) { const int code = P.ppadj[q], k = code >> 1, role = code & 1;
            const int other = fp_of_pose[role ? g.pp_i[k] : g.pp_j[k]];
            if (other >= 0) fn(Inc{other, k, role}); }
        for (int s = P.pl_start[p]; s < P.pl_start[p + 1]; ++s) { const int k = P.pl_order[s], fl = fl_of_lm[g.pl_l[k]];
            if (fl >= 0) fn(Inc{nfp + fl, k, 2}); }
    }
    void build_adjacency(const Plan &P, const std::vector<int32_t> &lm_k) {
        plan_ = &P; lm_k_ = &lm_k;
        lazy_pose.assign(nfp, 0); lazy_lm.assign(nfl, 0);
        if (opt.world > 1) {                                          // (a range test: window() is a 64-bit division per pose)
            const int W = opt.world, lo = (int)(((int64_t)opt.rank * nfp + W - 1) / W), hi = (int)(((int64_t)(opt.rank + 1) * nfp + W - 1) / W);
            std::fill(lazy_pose.begin(), lazy_pose.end(), (uint8_t)1);
            std::fill(lazy_pose.begin() + std::min(lo, nfp), lazy_pose.begin() + std::min(hi, nfp), (uint8_t)0); }
        std::vector<int32_t> obs_cnt(nfl, 0);
        big_assign(inc_start, (size_t)nv + 1, 0);
        // (the poses with a stored list — all of them, or a shard's own window: an eighth of the range at world 8 — and the landmarks are two
        // regions of their own: split evenly over ALL vertices, two of sixteen threads did a shard's pose work)
        int own_lo = 0, own_hi = nfp;
        if (opt.world > 1) { const int W = opt.world; own_lo = std::min(nfp, (int)(((int64_t)opt.rank * nfp + W - 1) / W)); own_hi = std::min(nfp, (int)(((int64_t)(opt.rank + 1) * nfp + W - 1) / W)); }
        auto both = [&](auto &&body) { parallel_chunks(own_hi - own_lo, 2048, [&](int64_t b0, int64_t e0, int) { body(own_lo + (int)b0, own_lo + (int)e0); });
                                       parallel_chunks(nfl, 2048, [&](int64_t b0, int64_t e0, int) { body(nfp + (int)b0, nfp + (int)e0); }); };
        both([&](int v0, int v1) {
            for (int v = v0; v < v1; ++v) { int n = 0;
                if (v < nfp && lazy_pose[v]) { }
                else if (v < nfp) { const int p = pose_of_fp[v];
                    for (int q = P.ppadj_start[p]; q < P.ppadj_start[p + 1]; ++q) { const int code = P.ppadj[q], k = code >> 1;
                        n += fp_of_pose[(code & 1) ? g.pp_i[k] : g.pp_j[k]] >= 0; }
                    for (int s = P.pl_start[p]; s < P.pl_start[p + 1]; ++s) n += fl_of_lm[g.pl_l[P.pl_order[s]]] >= 0;
                } else { const int l = lm_of_fl[v - nfp]; int wlo = INT32_MAX, whi = -1;
                    for (int q = P.lm_start[l]; q < P.lm_start[l + 1]; ++q) { const int fp = fp_of_pose[g.pl_p[lm_k[q]]];
                        if (fp >= 0) { ++n; if (opt.world > 1) { const int w = window(fp); wlo = std::min(wlo, w); whi = std::max(whi, w); } } }
                    obs_cnt[v - nfp] = n;
                    if (opt.world > 1 && n > 0 && wlo == whi && wlo != opt.rank) { lazy_lm[v - nfp] = 1; n = 0; } }
                inc_start[v + 1] = n; } });
        parallel_prefix(inc_start.data(), nv);
        if (inc_cap < (size_t)inc_start[nv] + 1) { inc_cap = (size_t)inc_start[nv] + 1; inc_store.reset(new Inc[inc_cap]); hint_huge_pages(inc_store.get(), inc_cap * sizeof(Inc)); }
        inc = inc_store.get();
        cone_obs_start.assign(nfl + 1, 0);
        for (int l = 0; l < nfl; ++l) cone_obs_start[l + 1] = cone_obs_start[l] + obs_cnt[l];
        big_resize(cone_obs, (size_t)cone_obs_start[nfl]); obs_lo.assign(nfl, INT32_MAX); obs_hi.assign(nfl, -1);
        // an observation edge is named by its INSERTION index here (epos); the assembly records are translated to the device
        // layout once it exists
        both([&](int v0, int v1) {
            for (int v = v0; v < v1; ++v) { Inc *o = &inc[inc_start[v]];
                if (v < nfp && lazy_pose[v]) { }
                else if (v < nfp) { const int p = pose_of_fp[v];
                    for (int q = P.ppadj_start[p]; q < P.ppadj_start[p + 1]; ++q) { const int code = P.ppadj[q], k = code >> 1, role = code & 1;
                        const int other = fp_of_pose[role ? g.pp_i[k] : g.pp_j[k]];
                        if (other >= 0) *o++ = {other, k, role}; }
                    for (int s = P.pl_start[p]; s < P.pl_start[p + 1]; ++s) { const int k = P.pl_order[s], fl = fl_of_lm[g.pl_l[k]];
                        if (fl >= 0) *o++ = {nfp + fl, k, 2}; }
                } else { const int lf = v - nfp, l = lm_of_fl[lf]; int32_t *co = &cone_obs[cone_obs_start[lf]]; const bool keep = !lazy_lm[lf];
                    for (int q = P.lm_start[l]; q < P.lm_start[l + 1]; ++q) { const int k = lm_k[q]; const int fp = fp_of_pose[g.pl_p[k]];
                        if (fp >= 0) { if (keep) *o++ = {fp, k, 3}; *co++ = fp; } }
                    int32_t *c0 = &cone_obs[cone_obs_start[lf]];
                    if (!std::is_sorted(c0, co)) std::sort(c0, co);
                    obs_lo[lf] = co > c0 ? c0[0] : INT32_MAX; obs_hi[lf] = co > c0 ? co[-1] : -1; }    // landmark observer lists (free-pose positions, ascending; edges grouped by pose arrive in that order)
            } });
    }

    bool has_observer(int l, int lo, int hi) const {   // any unassigned observer position in [lo, hi)
        // (most questions are settled by the landmark's first and last observer: 1.3 MB that stay in cache, where the lists are 25 MB at 800k poses)
        const int o0 = obs_lo[l], o1 = obs_hi[l];
        if (o1 < lo || o0 >= hi) return false;
        if (o0 >= lo && !assigned[o0]) return true;
        if (o1 < hi && !assigned[o1]) return true;
        auto b = cone_obs.begin() + cone_obs_start[l], e = cone_obs.begin() + cone_obs_start[l + 1];
        for (auto it = std::lower_bound(b, e, lo); it != e && *it < hi; ++it) if (!assigned[*it]) return true;
        return false;
    }

    // supernodes in elimination order, flat: the vertices of all of them in one array + offsets (18 000 little vectors at 100k poses — allocated one by
    // one in the recursion, moved list to list where its halves meet, freed one by one before the next plan — were ~1.5 ms of a 15 ms structure phase)
    struct SnList {
        std::vector<int32_t> v; std::vector<int64_t> off{0};
        struct View { const int32_t *b, *e;
            const int32_t *begin() const { return b; } const int32_t *end() const { return e; }
            size_t size() const { return (size_t)(e - b); } bool empty() const { return b == e; }
            int32_t operator[](size_t i) const { return b[i]; } };
        size_t size() const { return off.size() - 1; }
        bool empty() const { return off.size() == 1; }
        View operator[](size_t s) const { return View{v.data() + off[s], v.data() + off[s + 1]}; }
        View back() const { return (*this)[size() - 1]; }
        void push(const std::vector<int32_t> &verts) { v.insert(v.end(), verts.begin(), verts.end()); off.push_back((int64_t)v.size()); }
        void append(const SnList &o) { const int64_t base = (int64_t)v.size(); v.insert(v.end(), o.v.begin(), o.v.end());
            off.reserve(off.size() + o.size()); for (size_t s = 1; s < o.off.size(); ++s) off.push_back(base + o.off[s]); }
        void clear() { v.clear(); off.assign(1, 0); }
    };
    static void emit(SnList &out, const std::vector<int32_t> &verts) { if (!verts.empty()) out.push(verts); }
    SnList sn;                                       // supernodes (vertex lists) in elimination order

    // Multi-way split at the bottom of the tree.  Binary dissection down to the leaves leaves three levels of
    // separators that are ONE pose each (3 pivots: the cones around them are seen from far outside such a short range
    // and belong to separators higher up) — 14 336 of cfg4's 32 767 fronts, each paying a whole front's latency (child
    // gather, one panel, update matrix out) for three pivots.  A range of up to `cluster_ways` leaves is therefore cut
    // by all its split poses AT ONCE: they form one separator supernode (3 (p - 1) pivots) whose p children are the
    // leaves.  Bounded so that neither the cluster front nor a leaf can exceed the 63 scalars of the wave-per-front
    // kernels: rows <= separator poses + cones alive in the range + the two poses outside it.
    bool nd_multi(int a, int b, int un, std::vector<int32_t> &cones, SnList &out, int depth, bool in_own) {
        const int ways = opt.cluster_ways;
        if (ways <= 2) return false;
        int p = std::min(ways, (un + 1 + opt.leaf_poses) / (opt.leaf_poses + 1));       // ceil((un + 1) / (leaf + 1)) parts
        if (p <= 2 || un > ways * (opt.leaf_poses + 1) - 1) return false;
        std::vector<int32_t> un_pos; un_pos.reserve(un);
        for (int i = a; i < b; ++i) if (!assigned[i]) un_pos.push_back(i);
        // boundary of the cluster front = what the subtree (these poses + the cones alive here) touches outside itself:
        // poses across pose-pose edges (the separators of the ranges around it) and cones that already belong to a
        // separator higher up.  Exact, whatever p is.
        int nb = 0;
        { std::vector<int32_t> seen_p, seen_c;
          auto alive = [&](int l) { return std::find(cones.begin(), cones.end(), l) != cones.end(); };
          for (int i : un_pos) for_inc(i, [&](const Inc &e) { const int o = e.other;
              if (e.kind <= 1) { if ((o < a || o >= b || assigned[o]) && std::find(seen_p.begin(), seen_p.end(), o) == seen_p.end()) seen_p.push_back(o); }
              else { const int l = o - nfp; if (std::find(seen_c.begin(), seen_c.end(), l) == seen_c.end()) seen_c.push_back(l); } });
          int bc = 0; for (int l : seen_c) bc += !alive(l);
          nb = 3 * (int)seen_p.size() + 2 * bc; }
        std::vector<int32_t> sep_poses, cut, sep_cones, orphans;
        std::vector<std::vector<int32_t>> part_cones;
        // the cluster front must fit a wave (63 scalars) where that is possible at all; with 16-24 cones in view (frames of the
        // reference's coneMappingThreshold) the boundary alone is ~60 scalars, every front is a workgroup front anyway
        // and the bound becomes the 7-tile-row instance's 111 — otherwise such a range falls back to binary splits and
        // leaves level after level of ONE-pose separators (3 pivots each) with 60-100 boundary rows
        const int p_first = p;
        // the bounds tried in turn: (the cluster that holds the LAST pose: room for six appended keyframes,) room for two (grow_plan: an
        // appended pose becomes three boundary rows of the fronts between its neighbours and the root), a full wave, then the workgroup-front bound
        int limits[4], nlim = 0;
        { const int room = std::max(0, std::min(opt.grow_headroom, 30)), spine = std::max(room, std::min(opt.grow_spine_headroom, 30));
          if (b == nfp && spine > room) limits[nlim++] = 63 - spine;
          if (room > 0) limits[nlim++] = 63 - room;
          limits[nlim++] = 63;                                   // no room rather than a workgroup front
          if (opt.big_cluster_front > 63) limits[nlim++] = opt.big_cluster_front; }
        int li = 0, limit = limits[0];
        for (;; --p) {
            if (p <= 2) { if (++li < nlim) { limit = limits[li]; p = p_first + 1; continue; } return false; }
            sep_poses.clear(); cut.clear(); sep_cones.clear(); orphans.clear();
            // split poses: the unassigned ones at ranks (un * k) / p, k = 1 .. p - 1; part k = positions (cut[k], cut[k + 1])
            cut.push_back(a - 1);
            for (int k = 1; k < p; ++k) { const int m = un_pos[(size_t)((int64_t)un * k / p)];
                if (assigned[m]) continue; assigned[m] = 1; sep_poses.push_back(m); cut.push_back(m); }
            cut.push_back(b);
            const int np = (int)cut.size() - 1;
            auto part_of = [&](int i) { int k = 0; while (k + 1 < np && i > cut[k + 1]) ++k; return k; };
            // pose-pose edges that still span two parts pull their later endpoint into the separator
            for (int i : un_pos) { if (assigned[i]) continue;
                for_inc(i, [&](const Inc &e) { if (e.kind > 1) return;
                    const int j = e.other; if (j <= i || j < a || j >= b || assigned[j]) return;
                    if (part_of(j) != part_of(i)) { assigned[j] = 1; sep_poses.push_back(j); } }); }
            part_cones.assign(np, {});
            for (int l : cones) { int hit = -1, n = 0, k = 0;
                // one walk over the cone's observers inside the range (ascending: the part index only moves forward) instead of a search per part
                auto it = std::lower_bound(cone_obs.begin() + cone_obs_start[l], cone_obs.begin() + cone_obs_start[l + 1], a), e = cone_obs.begin() + cone_obs_start[l + 1];
                for (; it != e && *it < b && n < 2; ++it) { const int i = *it; if (assigned[i]) continue;      // (the split poses are assigned: no part)
                    while (k + 1 < np && i > cut[k + 1]) ++k;
                    if (k != hit) { hit = k; ++n; } }
                if (n >= 2) sep_cones.push_back(l); else if (n == 1) part_cones[hit].push_back(l); else orphans.push_back(l); }
            if (3 * (int)sep_poses.size() + 2 * (int)sep_cones.size() + nb <= limit) break;    // the cluster front fits a wave (or, second pass, a 7-tile-row workgroup)
            for (int m : sep_poses) assigned[m] = 0;                // too big: fewer parts (their larger pieces are dissected further)
        }
        const int np = (int)cut.size() - 1;
        cones.clear(); cones.shrink_to_fit();
        if (!orphans.empty()) { std::vector<int32_t> v; for (int l : orphans) v.push_back(nfp + l); emit(out, v); }
        for (int k = 0; k < np; ++k) nd(cut[k] + 1, cut[k + 1], part_cones[k], out, depth + 1, in_own);
        std::vector<int32_t> verts(sep_poses.begin(), sep_poses.end());
        for (int l : sep_cones) verts.push_back(nfp + l);
        emit(out, verts);
        return true;
    }

    // ---- pose-window shards, the top of the tree by WINDOWS (round 4).  Which windows see a landmark is one pass over the edges (two bit
    // masks per landmark); with them the splits between windows — a window's first pose and the cones seen from both sides — and the boundary
    // of another rank's window need neither that window's observer lists nor a walk over its edges.  Taken when the poses of different windows
    // are tied by the odometry chain only (an edge between two windows ends in the later window's first pose); any other graph takes the general
    // recursion below, which gives the same plan wherever its middle pose is a window's first one (windows of equal size, a power of two of them).
    bool by_window = false;
    std::vector<int32_t> wf;                         // wf[w] = first free-pose position of window w; wf[world] = nfp
    std::vector<uint64_t> seen_nb, seen_b;           // per free landmark: windows with an observer that is not / that is the window's first pose (w >= 1)
    std::vector<int32_t> opaque_of_pose;             // first pose position of an opaque supernode -> its window (-1); (a supernode's index in its list is not stable: the lists of the halves are concatenated)
    std::vector<uint64_t> pp_touch;                  // [world] bit x: the first pose of window x has an odometry edge into window w's interior
    static uint64_t wbits(int lo, int hi) { return hi <= lo ? 0 : ((hi >= 64 ? ~0ull : ((1ull << hi) - 1)) & ~((1ull << lo) - 1)); }
    void nd_top(int w0, int w1, std::vector<int32_t> &cones, SnList &out, int depth) {
        const int a = w0 == 0 ? 0 : wf[w0] + 1, b = wf[w1];          // the range's poses: behind the first pose of w0 (a separator higher up) up to the next window's
        if (w1 - w0 == 1) {
            if (w0 == opt.rank) { const size_t n0 = out.size(); nd(a, b, cones, out, depth, true); mark_own_range(out, n0); return; }
            // (nothing inside another rank's window has been assigned: only the windows' first poses are separators up here)
            const int np_ = std::max(b - a, 0);
            std::vector<int32_t> verts((size_t)np_ + cones.size());
            for (int i = 0; i < np_; ++i) verts[(size_t)i] = a + i;
            if (np_ > 0) { std::fill(assigned.begin() + a, assigned.begin() + b, (uint8_t)1); std::fill(opaque_pose.begin() + a, opaque_pose.begin() + b, (uint8_t)1);
                opaque_of_pose[a] = w0; }
            for (size_t c = 0; c < cones.size(); ++c) verts[(size_t)np_ + c] = nfp + cones[c];
            emit(out, verts);
            return; }
        const int wm = (w0 + w1 + 1) / 2, m = wf[wm];
        std::vector<int32_t> sep_poses;
        if (m < b && !assigned[m]) { assigned[m] = 1; sep_poses.push_back(m); }
        std::vector<int32_t> left, right, sep_cones, orphans;
        const uint64_t nl = wbits(w0, wm), bl = wbits(w0 + 1, wm), nr = wbits(wm, w1), br = wbits(wm + 1, w1);
        for (int l : cones) {
            const bool hl = ((seen_nb[l] & nl) | (seen_b[l] & bl)) != 0, hr = ((seen_nb[l] & nr) | (seen_b[l] & br)) != 0;
            if (hl && hr) sep_cones.push_back(l);
            else if (hl) left.push_back(l);
            else if (hr) right.push_back(l);
            else orphans.push_back(l);
        }
        cones.clear(); cones.shrink_to_fit();
        if (!orphans.empty()) { std::vector<int32_t> v; for (int l : orphans) v.push_back(nfp + l); emit(out, v); }
        const bool mine_l = opt.rank >= w0 && opt.rank < wm, mine_r = opt.rank >= wm && opt.rank < w1;
        if (mine_l == mine_r && (1 << depth) < host_threads()) {      // (neither half holds this rank's window: both are cheap; kept parallel for symmetry with the general path)
            SnList lo; std::thread th([&] { nd_top(w0, wm, left, lo, depth + 1); });
            SnList hi; nd_top(wm, w1, right, hi, depth + 1);
            th.join();
            out.append(lo); out.append(hi);
        } else { nd_top(w0, wm, left, out, depth); nd_top(wm, w1, right, out, depth); }
        std::vector<int32_t> verts(sep_poses.begin(), sep_poses.end());
        for (int l : sep_cones) verts.push_back(nfp + l);
        emit(out, verts);
    }

    // nested dissection over free-pose positions [a, b); `cones` = free landmarks alive in this range
    // Pose-window shards: the ranges of THIS rank's window that another rank sees as one opaque supernode (first and last supernode of the range,
    // named by their first vertex).  The shared top must come out the same on every rank: the owner therefore hands the shared fronts the
    // boundary the others compute for the range as a whole — the union over its supernodes — and not only what each of its subtrees carries up
    // (a window that falls apart, e.g. at a fixed pose between cones that all sit in separators, has several roots with smaller boundaries
    // hanging under different shared fronts: ranks then disagreed on the rows of the shared fronts).
    std::vector<std::pair<int32_t, int32_t>> own_ranges; std::mutex own_mu;
    void mark_own_range(const SnList &out, size_t n0) { if (out.size() > n0) { std::lock_guard<std::mutex> lk(own_mu); own_ranges.push_back({out[n0][0], out.back()[0]}); } }
    void nd(int a, int b, std::vector<int32_t> &cones, SnList &out, int depth, bool in_own = false) {
        int un = 0;
        for (int i = a; i < b; ++i) un += !assigned[i];
        if (opt.world > 1 && !in_own && un > 0) {                     // (the general recursion: a range that lies in this rank's window alone)
            int first = -1, last = -1;
            for (int i = a; i < b; ++i) if (!assigned[i]) { if (first < 0) first = i; last = i; }
            if (window(first) == window(last) && window(first) == opt.rank) { const size_t n0 = out.size(); nd(a, b, cones, out, depth, true); mark_own_range(out, n0); return; } }
        // pose-window shards: a range that lies in ONE window of ANOTHER rank is that rank's business — here it stays one opaque
        // supernode (its poses + the cones alive in it): the symbolic factorisation gives it the boundary the owner's whole
        // subtree has (the rows of the shared fronts above are the same on every rank), nothing below it is planned, stored or
        // uploaded on this rank.  A rank plans its own window and the shared top: 1 / world of the dissection.
        if (opt.world > 1 && un > 0) {
            int first = -1, last = -1;
            for (int i = a; i < b; ++i) if (!assigned[i]) { if (first < 0) first = i; last = i; }
            if (window(first) == window(last) && window(first) != opt.rank) {
                std::vector<int32_t> verts; verts.reserve((size_t)un + 2 * cones.size());
                for (int i = a; i < b; ++i) if (!assigned[i]) { verts.push_back(i); assigned[i] = 1; opaque_pose[i] = 1; }
                for (int l : cones) verts.push_back(nfp + l);
                emit(out, verts);
                return; } }
        if (un <= opt.leaf_poses) {
            std::vector<int32_t> verts;
            for (int i = a; i < b; ++i) if (!assigned[i]) { verts.push_back(i); assigned[i] = 1; }
            for (int l : cones) verts.push_back(nfp + l);
            emit(out, verts);
            return;
        }
        if (nd_multi(a, b, un, cones, out, depth, in_own)) return;
        // split pose: the middle unassigned one
        int m = -1, seen = 0;
        for (int i = a; i < b; ++i) if (!assigned[i]) { if (seen == un / 2) { m = i; break; } ++seen; }
        std::vector<int32_t> sep_poses{m};
        assigned[m] = 1;
        // pose-pose edges that still span the split pull their far endpoint into the separator
        for (int i = a; i < m; ++i) { if (assigned[i]) continue;
            for_inc(i, [&](const Inc &e) { if (e.kind > 1) return;
                const int j = e.other; if (j > m && j < b && !assigned[j]) { assigned[j] = 1; sep_poses.push_back(j); } }); }
        std::vector<int32_t> left, right, sep_cones, orphans;
        for (int l : cones) {
            bool hl = has_observer(l, a, m), hr = has_observer(l, m + 1, b);
            if (hl && hr) sep_cones.push_back(l);
            else if (hl) left.push_back(l);
            else if (hr) right.push_back(l);
            else orphans.push_back(l);
        }
        cones.clear(); cones.shrink_to_fit();
        if (!orphans.empty()) { std::vector<int32_t> v; for (int l : orphans) v.push_back(nfp + l); emit(out, v); }
        // the two halves touch disjoint pose ranges (and only read the shared tables): the top levels of the recursion
        // run them on separate host threads, each into its own list, concatenated in elimination order
        // (pose-window shards: a half that lies in one window of another rank is an opaque supernode at once — no work to share; the other
        // half keeps this level's thread budget, so a rank's own window is dissected on as many threads as a whole graph's)
        auto foreign = [&](int x, int y) { return opt.world > 1 && y > x && window(x) == window(y - 1) && window(x) != opt.rank; };
        const bool one_sided = foreign(a, m) || foreign(m + 1, b);
        if (!one_sided && (1 << depth) < host_threads() && un > 2048) {
            SnList lo;
            std::thread th([&] { nd(a, m, left, lo, depth + 1, in_own); });
            SnList hi; nd(m + 1, b, right, hi, depth + 1, in_own);
            th.join();
            out.append(lo); out.append(hi);
        } else { const int dn = one_sided ? depth : depth + 1; nd(a, m, left, out, dn, in_own); nd(m + 1, b, right, out, dn, in_own); }
        std::vector<int32_t> verts(sep_poses.begin(), sep_poses.end());
        for (int l : sep_cones) verts.push_back(nfp + l);
        emit(out, verts);
    }
};

}  // namespace

namespace {
struct PlanScratch { std::unique_ptr<Builder> B; std::vector<int32_t> lm_k, sn_of, vpos, gidx, parent, stamp, asm_n; std::vector<std::vector<int32_t>> bndv, kids; };
}
bool build_plan(const HostGraph &g, const PlanOptions &opt_in, Plan &plan, std::string &err, std::shared_ptr<void> *workspace) {
    auto t0 = std::chrono::steady_clock::now();
    const bool pt_on = opt_in.timing; auto pt_prev = t0;              // gs_debug_options.plan_timing: phase times on stderr
#define GS_PT(i) do { if (pt_on) { auto n_ = std::chrono::steady_clock::now(); std::fprintf(stderr, "plan phase %d: %.2f ms\n", (i), std::chrono::duration<double, std::milli>(n_ - pt_prev).count()); pt_prev = n_; } } while (0)
    // A handle that plans again keeps its memory: the arrays of the previous plan are emptied, not freed (their pages are mapped already —
    // freeing and re-allocating ~60 MB at 100k poses, ~200 MB for a pose-window shard of 800k, was up to a third of a structure phase), and so
    // is the scratch of this function (adjacency, supernode lists, boundaries) when the caller lends a workspace.
    { Plan old = std::move(plan); plan = Plan();
#define GS_KEEP(m) do { plan.m = std::move(old.m); plan.m.clear(); } while (0)
#define GS_KEEP_SIZED(m) do { plan.m = std::move(old.m); } while (0)      /* overwritten in full by a parallel fill (pfill): the size stays, nothing is re-initialised */
      GS_KEEP_SIZED(pose_gidx); GS_KEEP_SIZED(lm_gidx); GS_KEEP_SIZED(pl_order); GS_KEEP(pp_order); GS_KEEP_SIZED(pl_start); GS_KEEP(lm_start); GS_KEEP(lm_edges); GS_KEEP_SIZED(ppadj_start); GS_KEEP(ppadj);
      GS_KEEP(ell_ins); GS_KEEP_SIZED(ell_of_ins); GS_KEEP(ppinc); GS_KEEP(wt_grp_start); GS_KEEP(wt_desc); GS_KEEP(grp_lm); GS_KEEP(grp_pos_start); GS_KEEP(grp_pos); GS_KEEP(ell_dst);
      GS_KEEP(lm_grp_start); GS_KEEP(grp_slot); GS_KEEP(fronts); GS_KEEP(bnd_rows); GS_KEEP(child_map); GS_KEEP(children); GS_KEEP(asm_recs); GS_KEEP(level_start); GS_KEEP(level_fronts);
      GS_KEEP_SIZED(pl_rank); GS_KEEP_SIZED(pp_rank); GS_KEEP_SIZED(pose_known); GS_KEEP_SIZED(lm_known); GS_KEEP(level_start_owned); GS_KEEP(level_fronts_owned); GS_KEEP(level_start_shared);
      GS_KEEP(level_fronts_shared); GS_KEEP(x_off);
#undef GS_KEEP
#undef GS_KEEP_SIZED
    }
    PlanOptions opt = opt_in;
    const bool leaf_auto = opt.leaf_poses <= 0;
    if (opt.leaf_poses <= 0) opt.leaf_poses = 8;
    if (opt.cluster_ways <= 0) opt.cluster_ways = 8;
    std::unique_ptr<PlanScratch> local_scratch; PlanScratch *scratch;
    if (workspace) { if (!*workspace) *workspace = std::shared_ptr<void>(new PlanScratch, [](void *p) { delete static_cast<PlanScratch *>(p); });
        scratch = static_cast<PlanScratch *>(workspace->get()); }
    else { local_scratch = std::make_unique<PlanScratch>(); scratch = local_scratch.get(); }
    { auto nb = std::make_unique<Builder>(g, opt); if (scratch->B) nb->adopt(*scratch->B); scratch->B = std::move(nb); }
    Builder &B = *scratch->B;
    B.index_vertices();
    if (B.nv == 0) { err = "no free vertex"; return false; }
    // Leaf size by the graph (round 4): the binary part of the dissection halves the pose range until a range fits one multi-way cluster
    // (ways x (leaf + 1) - 1 poses); a smaller leaf that does NOT add a binary level makes every leaf front smaller at the same tree depth —
    // at 100k poses leaves of 6 poses all fit 47 scalars (the three-tile-row leaf instance, 84 registers) where leaves of 8 reach 50: factor
    // 0.170 -> 0.160 ms, backward solve 0.067 -> 0.065 (+3 % iterations/s); at 1M poses 6 would add a level (-10 %) and 8 stays; at 10k
    // neutral (scripts/r4_e.sh).  Only when nobody asked for a size, and only for narrow views (wide views: workgroup fronts, measured best at 8).
    if (leaf_auto) {
        auto depth_of = [&](int leaf) { int64_t un = B.nfp, cap = (int64_t)opt.cluster_ways * (leaf + 1) - 1; int dd = 0; while (un > cap) { un = un / 2; ++dd; } return dd; };
        const int d8 = depth_of(8);
        const int best = depth_of(6) == d8 ? 6 : 8;                  // (7 keeps some leaves above 47 scalars: measured neutral at 100k, -2 % at 1M poses)
        B.opt.leaf_poses = opt.leaf_poses = best;
    }
    const int N = g.n_poses(), M = g.n_lms(), Epl = g.n_pl(), Epp = g.n_pp();

    GS_PT(0);
    // ---- edges grouped by pose and by landmark (insertion indices; the device layout of the observation edges follows the shard
    // assignment further down: a rank lays out the poses it sweeps only)
    // (room for grow_plan's appended runs is reserved BEFORE the arrays are filled: reserving afterwards re-allocated and copied ~60 MB at 100k poses)
    plan.pp_order.reserve((size_t)Epp + TAIL_PP);
    auto size_only = [](std::vector<int32_t> &v, size_t n, size_t room) { if (v.size() != n) { v.clear(); v.reserve(n + room); hint_huge(v); v.resize(n); } };      // (every entry is written below: a recycled array is not initialised again)
    size_only(plan.pl_order, (size_t)Epl, TAIL_PL); size_only(plan.pl_start, (size_t)N + 1, 0);
    plan.pp_order.resize(Epp);
    for (int k = 0; k < Epp; ++k) plan.pp_order[k] = k;
    // The reference adds a keyframe's observation edges behind its pose (src/slam.cpp:433-459, 525-550): the edges arrive grouped by pose.
    // Then the grouping is the identity and the ranges come from a scan — on all host threads; any other insertion order takes the counting sort.
    // A pose-window shard walks the edges of ALL windows here, and only here: ONE pass checks the order, writes the identity and the ranges and
    // collects, per landmark, which windows see it (Builder::nd_top) — these passes were 70 of a rank's 480 ms at 8 x 100k poses when they were
    // four and sequential, 3 of 31 as three parallel ones.
    B.by_window = false;
    const bool want_win = opt.world > 1 && opt.world <= 64 && B.nfp >= 4 * opt.world && opt.by_window;
    std::vector<uint8_t> win_of;
    bool chain_ok = false;
    const bool given = opt.lm_seen_interior != nullptr && opt.lm_seen_first != nullptr;      // rank-local ingestion: the masks come with the graph
    auto is_first = [&](int fp, int w) { return w >= 1 && fp == B.wf[w]; };
    if (want_win) {
        const int W = opt.world;
        B.wf.assign(W + 1, 0);
        for (int w = 0; w <= W; ++w) B.wf[w] = (int32_t)(((int64_t)w * B.nfp + W - 1) / W);
        win_of.resize(B.nfp);
        parallel_chunks(B.nfp, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t i = b0; i < e0; ++i) win_of[(size_t)i] = (uint8_t)B.window((int)i); });
        // odometry edges between windows must end in the later window's first pose
        std::vector<uint8_t> bad(host_threads() + 1, 0);
        B.pp_touch.assign(W, 0);
        std::vector<std::vector<uint64_t>> touch_t(host_threads() + 1, std::vector<uint64_t>(W, 0));
        parallel_chunks(Epp, 16384, [&](int64_t b0, int64_t e0, int t) { auto &tt = touch_t[t];
            for (int64_t k = b0; k < e0; ++k) { const int fi = B.fp_of_pose[g.pp_i[(size_t)k]], fj = B.fp_of_pose[g.pp_j[(size_t)k]];
                if (fi < 0 || fj < 0) continue;
                const int wi = win_of[fi], wj = win_of[fj]; const bool bi = is_first(fi, wi), bj = is_first(fj, wj);
                if (!bi && !bj) { if (wi != wj) bad[t] = 1; continue; }
                if (bi && !bj) tt[wj] |= 1ull << wi;
                if (bj && !bi) tt[wi] |= 1ull << wj; } });
        chain_ok = true; for (uint8_t v : bad) chain_ok = chain_ok && !v;
        if (chain_ok) { for (auto &tt : touch_t) for (int w = 0; w < W; ++w) B.pp_touch[w] |= tt[w];
            B.seen_nb.assign(B.nfl, 0); B.seen_b.assign(B.nfl, 0);
            if (given) for (int fl = 0; fl < B.nfl; ++fl) { B.seen_nb[fl] = opt.lm_seen_interior[B.lm_of_fl[fl]]; B.seen_b[fl] = opt.lm_seen_first[B.lm_of_fl[fl]]; } } }
    if (given && !(want_win && chain_ok)) { err = "landmark windows were handed over (gs_dist_set_landmark_windows), but the graph cannot be planned by windows: more than 64 ranks, fewer than 4 free poses per window, or odometry edges between the interiors of two windows"; return false; }
    bool by_pose = true;
    { std::vector<uint8_t> bad(host_threads() + 1, 0), miss(host_threads() + 1, 0);
      parallel_chunks(Epl, 16384, [&](int64_t b0, int64_t e0, int t) {
          int32_t prev = b0 > 0 ? g.pl_p[(size_t)b0 - 1] : -1;
          for (int64_t k = b0; k < e0; ++k) { const int32_t p = g.pl_p[(size_t)k];
              plan.pl_order[(size_t)k] = (int32_t)k;
              if (p < prev) bad[t] = 1;
              for (int32_t q = prev + 1; q <= p; ++q) plan.pl_start[(size_t)q] = (int32_t)k;      // (out of order: nothing is written; the counting sort below redoes all of it)
              prev = p;
              if (chain_ok) { const int fp = B.fp_of_pose[(size_t)p]; if (fp < 0) continue;
                  const int fl = B.fl_of_lm[g.pl_l[(size_t)k]]; if (fl < 0) continue;
                  const int w = win_of[fp]; uint64_t *tgt = is_first(fp, w) ? &B.seen_b[fl] : &B.seen_nb[fl]; const uint64_t bit = 1ull << w;
                  if (given) { if (!(*tgt & bit)) miss[t] = 1; }      // (handed over: the edges that ARE here must agree with them)
                  else if (!(__atomic_load_n(tgt, __ATOMIC_RELAXED) & bit)) __atomic_fetch_or(tgt, bit, __ATOMIC_RELAXED); } } });
      for (uint8_t b : bad) by_pose = by_pose && !b;
      for (uint8_t b : miss) if (b) { err = "an observation edge of this graph is missing from the landmark windows handed over (gs_dist_set_landmark_windows)"; return false; } }
    if (given && !by_pose) { err = "landmark windows were handed over, but the observation edges do not arrive grouped by pose"; return false; }
    if (by_pose) { for (int32_t q = (Epl > 0 ? g.pl_p[(size_t)Epl - 1] + 1 : 0); q <= N; ++q) plan.pl_start[(size_t)q] = Epl; }
    else {
        std::fill(plan.pl_start.begin(), plan.pl_start.end(), 0);
        for (int k = 0; k < Epl; ++k) plan.pl_start[g.pl_p[k] + 1]++;
        for (int p = 0; p < N; ++p) plan.pl_start[p + 1] += plan.pl_start[p];
        std::vector<int32_t> fill(plan.pl_start.begin(), plan.pl_start.end() - 1);
        for (int k = 0; k < Epl; ++k) plan.pl_order[fill[g.pl_p[k]]++] = k; }
    // ---- pose-window shards: which windows see a landmark (Builder::nd_top)
    if (want_win && chain_ok && by_pose) { B.opaque_of_pose.assign(B.nfp, -1); B.by_window = true; }
    // by windows: the poses whose edges this rank's plan is built from — its own window, every window's first pose (the separators of the
    // shared top) and the fixed poses; everything else of the other windows is summarised by the masks above
    std::vector<int32_t> ing;                                         // insertion indices, ascending
    std::vector<uint8_t> ing_flag;
    if (B.by_window) { const int W = opt.world;
        ing_flag.assign(N, 0);
        parallel_chunks(N, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t p = b0; p < e0; ++p) { const int fp = B.fp_of_pose[(size_t)p];
            ing_flag[(size_t)p] = fp < 0 || (fp >= B.wf[opt.rank] && fp < B.wf[opt.rank + 1]); } });
        for (int x = 1; x < W; ++x) if (B.wf[x] < B.nfp) ing_flag[B.pose_of_fp[B.wf[x]]] = 1;
        { const int T = chunk_count(N, 16384); std::vector<int32_t> cnt((size_t)T + 1, 0);      // (flags -> ascending list: counts per part, then every part writes its own)
          parallel_chunks(N, 16384, [&](int64_t b0, int64_t e0, int t) { int32_t c = 0; for (int64_t p = b0; p < e0; ++p) c += ing_flag[(size_t)p]; cnt[(size_t)t + 1] = c; });
          for (int t = 0; t < T; ++t) cnt[(size_t)t + 1] += cnt[(size_t)t];
          ing.resize((size_t)cnt[(size_t)T]);
          parallel_chunks(N, 16384, [&](int64_t b0, int64_t e0, int t) { int32_t c = cnt[(size_t)t]; for (int64_t p = b0; p < e0; ++p) if (ing_flag[(size_t)p]) ing[(size_t)c++] = (int32_t)p; }); } }
    GS_PT(2);
    // landmark -> its edges (insertion indices, pose order); turned into ELL indices at the end (lm_edges, single GPU only).  A stable
    // counting sort by landmark over the pose-grouped sequence, in chunks: per chunk a histogram, offsets per (chunk, landmark), scatter.
    plan.lm_start.assign(M + 1, 0);
    std::vector<int32_t> &lm_k = scratch->lm_k;
    if (B.by_window) {                                                // the edges of the poses in `ing` only (pose order = insertion order here)
        // (the same stable counting sort as below, in chunks of `ing`)
        const int64_t U = (int64_t)ing.size();
        const int C = (int64_t)chunk_count(U, 8192) * M <= ((int64_t)1 << 26) ? chunk_count(U, 8192) : 1;
        std::vector<std::vector<int32_t>> cnt(C);
        auto lo = [&](int c) { return U * c / C; };
        parallel_chunks(C, 1, [&](int64_t c0, int64_t c1, int) { for (int c = (int)c0; c < (int)c1; ++c) { auto &h = cnt[c]; h.assign((size_t)M, 0);
            for (int64_t u = lo(c); u < lo(c + 1); ++u) { const int p = ing[(size_t)u]; for (int q = plan.pl_start[p]; q < plan.pl_start[p + 1]; ++q) h[g.pl_l[q]]++; } } });
        parallel_chunks(M, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t l = b0; l < e0; ++l) { int32_t n = 0;
            for (int c = 0; c < C; ++c) { const int32_t v = cnt[c][(size_t)l]; cnt[c][(size_t)l] = n; n += v; }
            plan.lm_start[(size_t)l + 1] = n; } });
        for (int l = 0; l < M; ++l) plan.lm_start[l + 1] += plan.lm_start[l];
        big_resize(lm_k, (size_t)plan.lm_start[M]);
        parallel_chunks(C, 1, [&](int64_t c0, int64_t c1, int) { for (int c = (int)c0; c < (int)c1; ++c) { auto &h = cnt[c];
            for (int64_t u = lo(c); u < lo(c + 1); ++u) { const int p = ing[(size_t)u];
                for (int q = plan.pl_start[p]; q < plan.pl_start[p + 1]; ++q) { const int l = g.pl_l[q]; lm_k[(size_t)plan.lm_start[l] + h[l]++] = q; } } } });
    } else {
    big_resize(lm_k, (size_t)Epl);
    { const int C = (int64_t)chunk_count(Epl, 1 << 18) * M <= ((int64_t)1 << 26) ? chunk_count(Epl, 1 << 18) : 1;
      std::vector<std::vector<int32_t>> cnt(C);
      auto lo = [&](int c) { return (int64_t)Epl * c / C; };
      parallel_chunks(C, 1, [&](int64_t c0, int64_t c1, int) { for (int c = (int)c0; c < (int)c1; ++c) { auto &h = cnt[c]; h.assign((size_t)M, 0);
          for (int64_t pos = lo(c); pos < lo(c + 1); ++pos) h[g.pl_l[plan.pl_order[(size_t)pos]]]++; } });
      parallel_chunks(M, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t l = b0; l < e0; ++l) { int32_t n = 0;
          for (int c = 0; c < C; ++c) { const int32_t v = cnt[c][(size_t)l]; cnt[c][(size_t)l] = n; n += v; }      // offset of chunk c inside landmark l's run
          plan.lm_start[(size_t)l + 1] = n; } });
      for (int l = 0; l < M; ++l) plan.lm_start[l + 1] += plan.lm_start[l];
      parallel_chunks(C, 1, [&](int64_t c0, int64_t c1, int) { for (int c = (int)c0; c < (int)c1; ++c) { auto &h = cnt[c];
          for (int64_t pos = lo(c); pos < lo(c + 1); ++pos) { const int k = plan.pl_order[(size_t)pos]; const int l = g.pl_l[k]; lm_k[(size_t)plan.lm_start[l] + h[l]++] = k; } } }); }
    }
    // pose -> incident pp edges, and the flattened incidence records {edge, role, i, j}
    pfill(plan.ppadj_start, (size_t)N + 1, (int32_t)0);
    if (B.by_window) {                                                // incidences of the poses in `ing` only: the edges that touch one, picked on the host threads
        const int T = host_threads() + 1; std::vector<std::vector<int32_t>> pick(T);
        parallel_chunks(Epp, 16384, [&](int64_t b0, int64_t e0, int t) { for (int64_t k = b0; k < e0; ++k) if (ing_flag[g.pp_i[(size_t)k]] | ing_flag[g.pp_j[(size_t)k]]) pick[t].push_back((int32_t)k); });
        for (auto &v : pick) for (int k : v) { if (ing_flag[g.pp_i[k]]) plan.ppadj_start[g.pp_i[k] + 1]++; if (ing_flag[g.pp_j[k]]) plan.ppadj_start[g.pp_j[k] + 1]++; }
        parallel_prefix(plan.ppadj_start.data(), N);
        plan.ppadj.resize((size_t)plan.ppadj_start[N]);
        std::vector<int32_t> fill(plan.ppadj_start.begin(), plan.ppadj_start.end() - 1);
        for (auto &v : pick) for (int k : v) { if (ing_flag[g.pp_i[k]]) plan.ppadj[fill[g.pp_i[k]]++] = 2 * k; if (ing_flag[g.pp_j[k]]) plan.ppadj[fill[g.pp_j[k]]++] = 2 * k + 1; }
    } else {
    for (int k = 0; k < Epp; ++k) { plan.ppadj_start[g.pp_i[k] + 1]++; plan.ppadj_start[g.pp_j[k] + 1]++; }
    for (int p = 0; p < N; ++p) plan.ppadj_start[p + 1] += plan.ppadj_start[p];
    big_resize(plan.ppadj, 2 * (size_t)Epp);
    { std::vector<int32_t> fill(plan.ppadj_start.begin(), plan.ppadj_start.end() - 1);
      for (int k = 0; k < Epp; ++k) { plan.ppadj[fill[g.pp_i[k]]++] = 2 * k; plan.ppadj[fill[g.pp_j[k]]++] = 2 * k + 1; } }
    }
    big_resize(plan.ppinc, plan.ppadj.size() * 4);
    parallel_chunks((int64_t)plan.ppadj.size(), 16384, [&](int64_t b0, int64_t e0, int) {
        for (size_t q = (size_t)b0; q < (size_t)e0; ++q) { const int code = plan.ppadj[q], k = code >> 1;
            plan.ppinc[4 * q] = k; plan.ppinc[4 * q + 1] = code & 1; plan.ppinc[4 * q + 2] = g.pp_i[k]; plan.ppinc[4 * q + 3] = g.pp_j[k]; } });

    GS_PT(1);
    // room to grow (grow_plan) only where fronts fit a wave anyway: with more than ~10 cones in view the cluster fronts are workgroup
    // fronts, such a plan cannot grow, and keeping them below 57 would only cost fronts (K = 16: 26 571 instead of 21 026, -5 % it/s)
    int kmax_all = 0;                                                 // most observation edges at one pose
    { std::vector<int> kt(host_threads() + 1, 0);
      parallel_chunks(N, 16384, [&](int64_t b0, int64_t e0, int t) { int m = 0; for (int64_t p = b0; p < e0; ++p) m = std::max(m, plan.pl_start[(size_t)p + 1] - plan.pl_start[(size_t)p]); kt[t] = m; });
      for (int m : kt) kmax_all = std::max(kmax_all, m); }
    { const int kmax0 = kmax_all;
      if (kmax0 > 10 && leaf_auto) B.opt.leaf_poses = 8;           // (wide views keep leaves of 8 poses)
      if (kmax0 > 10) B.opt.grow_headroom = B.opt.grow_spine_headroom = 0;
      if (B.opt.big_cluster_front < 0) B.opt.big_cluster_front = kmax0 > 10 ? 111 : 0; }
    // ---- elimination order by nested dissection ----
    B.build_adjacency(plan, lm_k);
    GS_PT(21);
    B.assigned.assign(B.nfp, 0); B.opaque_pose.assign(B.nfp, 0);
    { std::vector<int32_t> all(B.nfl); for (int l = 0; l < B.nfl; ++l) all[l] = l;
      if (B.by_window) B.nd_top(0, opt.world, all, B.sn, 0); else B.nd(0, B.nfp, all, B.sn, 0); }
    const int S = (int)B.sn.size();
    std::vector<int32_t> &sn_of = scratch->sn_of, &vpos = scratch->vpos, &gidx = scratch->gidx;
    size_only(sn_of, (size_t)B.nv, 0); size_only(vpos, (size_t)B.nv, 0); size_only(gidx, (size_t)B.nv, 0);      // (written in full just below)
    // the supernode lists are one flat array in elimination order: a vertex's position is its index there, its first scalar the running sum of
    // the dimensions before it — supernodes in chunks on the host threads (a pose-window shard numbers 0.9 M vertices here)
    { const auto &sv = B.sn.v; const auto &so = B.sn.off;
      if ((int64_t)sv.size() != B.nv) { err = sv.size() < (size_t)B.nv ? "ordering lost a vertex" : "vertex emitted twice"; return false; }
      const int64_t nvv = B.nv; const int T = chunk_count(nvv, 16384);      // (by position, not by supernode: another rank's window is ONE supernode of 110k vertices)
      std::vector<int64_t> sc0((size_t)T + 1, 0);
      parallel_chunks(nvv, 16384, [&](int64_t b0, int64_t e0, int t) { int64_t a = 0; for (int64_t j = b0; j < e0; ++j) a += B.dim(sv[(size_t)j]); sc0[(size_t)t + 1] = a; });
      for (int t = 0; t < T; ++t) sc0[(size_t)t + 1] += sc0[(size_t)t];
      parallel_chunks(nvv, 16384, [&](int64_t b0, int64_t e0, int t) { int32_t sc = (int32_t)sc0[(size_t)t];
          int64_t s2 = (std::upper_bound(so.begin(), so.end(), b0) - so.begin()) - 1;      // the supernode position b0 lies in
          for (int64_t j = b0; j < e0; ++j) { while (j >= so[(size_t)s2 + 1]) ++s2;
              const int v = sv[(size_t)j]; sn_of[v] = (int32_t)s2; vpos[v] = (int32_t)j; gidx[v] = sc; sc += B.dim(v); } });
      std::vector<uint8_t> twice(host_threads() + 1, 0);                // (nv entries for nv vertices: a vertex emitted twice leaves another one out — and one of its two places disagrees)
      parallel_chunks(B.nv, 16384, [&](int64_t b0, int64_t e0, int t) { for (int64_t j = b0; j < e0; ++j) if (vpos[sv[(size_t)j]] != (int32_t)j) twice[t] = 1; });
      for (uint8_t b : twice) if (b) { err = "vertex emitted twice"; return false; }
      plan.n_scalar = (int32_t)sc0[(size_t)T]; }
    plan.pose_gidx.reserve((size_t)N + TAIL_POSES); plan.lm_gidx.reserve((size_t)M + TAIL_LMS);
    pfill(plan.pose_gidx, (size_t)N, (int32_t)-1); pfill(plan.lm_gidx, (size_t)M, (int32_t)-1);
    parallel_chunks(B.nfp, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t i = b0; i < e0; ++i) plan.pose_gidx[B.pose_of_fp[(size_t)i]] = gidx[(size_t)i]; });
    parallel_chunks(B.nfl, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t l = b0; l < e0; ++l) plan.lm_gidx[B.lm_of_fl[(size_t)l]] = gidx[(size_t)B.nfp + (size_t)l]; });

    GS_PT(3);
    // ---- symbolic factorisation over supernodes ----
    // boundary of a supernode = the later-eliminated vertices its own vertices touch (independent per supernode: host
    // threads) + what its children's boundaries carry beyond it (one sequential bottom-up sweep, elimination order)
    std::vector<std::vector<int32_t>> &bndv = scratch->bndv, &kids = scratch->kids; bndv.resize(S); kids.resize(S);
    for (auto &v : bndv) v.clear();
    for (auto &v : kids) v.clear();                                   // (a workspace's lists keep their capacity)
    std::vector<int32_t> &parent = scratch->parent, &stamp = scratch->stamp; parent.assign(S, -1); stamp.assign(B.nv, -1);
    auto is_opaque = [&](int s) { return B.sn[s][0] < B.nfp && B.opaque_pose[B.sn[s][0]] != 0; };      // (an opaque supernode lists its poses first)
    parallel_chunks(S, 512, [&](int64_t b0, int64_t e0, int) {
        std::vector<int32_t> st(B.nv, -1);
        for (int s = (int)b0; s < (int)e0; ++s) { auto &bd = bndv[s];
            if (is_opaque(s)) continue;
            for (int v : B.sn[s]) B.for_inc(v, [&](const Builder::Inc &e) { const int w = e.other;
                if (sn_of[w] > s && st[w] != s) { st[w] = s; bd.push_back(w); } }); } });
    // another rank's window (one supernode of ~100k poses + its cones): its vertices in chunks on the host threads.  Nearly everything they
    // touch is inside; the few later-eliminated neighbours are collected per chunk (short lists, searched linearly) and merged — the order
    // does not matter, a boundary is sorted by elimination position below
    if (B.by_window) {
        // by windows: what another rank's window touches outside itself are the cones its poses see that it does not own (the masks) and the
        // first poses of windows that see one of its own cones or follow / precede it on the odometry chain — no walk over its edges
        // (a cone is "its own" iff it sits in the supernode: sn_of says so; the opaque supernodes one task each)
        std::vector<int32_t> ops; for (int s = 0; s < S; ++s) if (is_opaque(s)) ops.push_back(s);
        parallel_chunks((int64_t)ops.size(), 1, [&](int64_t o0, int64_t o1, int) { for (int64_t o = o0; o < o1; ++o) { const int s = ops[(size_t)o];
            const int w = B.opaque_of_pose[B.sn[s][0]]; auto &bd = bndv[s];
            uint64_t firsts = B.pp_touch[w];
            for (int v : B.sn[s]) if (v >= B.nfp) firsts |= B.seen_b[v - B.nfp];
            for (int x = 1; x < opt.world; ++x) if ((firsts >> x) & 1) { const int m = B.wf[x]; if (sn_of[m] > s) bd.push_back(m); }
            const uint64_t bit = 1ull << w;
            for (int l = 0; l < B.nfl; ++l) if ((B.seen_nb[l] & bit) && sn_of[B.nfp + l] > s) bd.push_back(B.nfp + l); } });
    } else
    for (int s = 0; s < S; ++s) if (is_opaque(s)) {
        const auto vs = B.sn[s]; const int C = chunk_count((int64_t)vs.size(), 8192);
        std::vector<std::vector<int32_t>> cand(C);
        parallel_chunks(C, 1, [&](int64_t c0, int64_t c1, int) { for (int c = (int)c0; c < (int)c1; ++c) { auto &cd = cand[c];
            for (size_t i = vs.size() * c / C; i < vs.size() * (c + 1) / C; ++i) B.for_inc(vs[i], [&](const Builder::Inc &e) { const int w = e.other;
                if (sn_of[w] > s && std::find(cd.begin(), cd.end(), w) == cd.end()) cd.push_back(w); }); } });
        auto &bd = bndv[s];
        for (auto &cd : cand) for (int w : cd) if (std::find(bd.begin(), bd.end(), w) == bd.end()) bd.push_back(w); }
    // this rank's own ranges as the other ranks see them: the union of what their supernodes touch beyond the range, handed to the shared front
    // that eliminates its first vertex like a child's boundary (no extend-add comes with it: the range's real roots keep their own parents,
    // whose rows hold everything a root carries — they lie on the path this union travels)
    std::vector<std::vector<std::vector<int32_t>>> virt;
    if (opt.world > 1 && !B.own_ranges.empty()) { virt.resize(S);
        for (auto &r : B.own_ranges) { const int s0 = sn_of[r.first], s1 = sn_of[r.second];
            std::vector<int32_t> U;
            for (int s = s0; s <= s1; ++s) for (int w : bndv[s]) if (sn_of[w] > s1 && stamp[w] != -2 - s1) { stamp[w] = -2 - s1; U.push_back(w); }
            if (U.empty()) continue;
            int first = U[0]; for (int w : U) if (vpos[w] < vpos[first]) first = w;
            virt[sn_of[first]].push_back(std::move(U)); }
        std::fill(stamp.begin(), stamp.end(), -1); }
    for (int s = 0; s < S; ++s) {
        auto &bd = bndv[s];
        const bool has_virt = !virt.empty() && !virt[s].empty();
        if (!kids[s].empty() || has_virt) {
            for (int w : bd) stamp[w] = s;
            for (int c : kids[s]) for (int w : bndv[c]) if (sn_of[w] != s && stamp[w] != s) { stamp[w] = s; bd.push_back(w); }
            if (has_virt) for (auto &U : virt[s]) for (int w : U) if (sn_of[w] != s && stamp[w] != s) { stamp[w] = s; bd.push_back(w); } }
        std::sort(bd.begin(), bd.end(), [&](int x, int y) { return vpos[x] < vpos[y]; });
        if (!bd.empty()) { parent[s] = sn_of[bd[0]]; kids[parent[s]].push_back(s); }
    }

    GS_PT(4);
    // ---- fronts ----  (three passes: sizes in parallel, offsets in one sweep, contents in parallel)
    plan.fronts.resize(S);
    std::vector<int32_t> &asm_n = scratch->asm_n; asm_n.resize(S);
    parallel_chunks(S, 512, [&](int64_t b0, int64_t e0, int) {
        for (int s = (int)b0; s < (int)e0; ++s) {
            Front &F = plan.fronts[s];
            F.parent = parent[s]; F.piv0 = gidx[B.sn[s][0]];
            for (int v : B.sn[s]) F.npiv += B.dim(v);
            for (int w : bndv[s]) F.nbnd += B.dim(w);
            F.opaque = (B.sn[s][0] < B.nfp && B.opaque_pose[B.sn[s][0]]) ? 1 : 0;     // (an opaque supernode lists its poses first)
            int n = 0;
            if (!F.opaque) for (int v : B.sn[s]) { ++n;
                B.for_inc(v, [&](const Builder::Inc &e) { n += vpos[e.other] > vpos[v]; }); }     // the earlier endpoint owns the block
            asm_n[s] = n;
        } });
    { int64_t nb = 0, nm = 0, na = 0;
      for (int s = 0; s < S; ++s) {
          Front &F = plan.fronts[s];
          F.bnd_off = nb; nb += F.nbnd;
          F.map_off = nm; nm += F.nbnd;                               // this front's boundary rows -> rows of its parent's front
          F.level = 0;
          for (int c : kids[s]) F.level = std::max(F.level, plan.fronts[c].level + 1);
          F.child_off = (int32_t)plan.children.size(); F.child_cnt = (int32_t)kids[s].size();
          for (int c : kids[s]) plan.children.push_back(c);
          F.asm_off = (int32_t)na; F.asm_cnt = asm_n[s]; na += asm_n[s];
          F.L_off = plan.l_doubles; F.U_off = plan.u_doubles;
          if (F.opaque) continue;                                     // another rank's subtree: no factor, no update matrix, no share in the sizes
          plan.max_front = std::max(plan.max_front, F.npiv + F.nbnd);
          plan.l_doubles += (int64_t)(F.npiv + F.nbnd + 1) * F.npiv;
          plan.u_doubles += (int64_t)(F.nbnd + 1) * F.nbnd;
          for (int k = 0; k < F.npiv; ++k) { int64_t r2 = F.npiv + F.nbnd + 1 - k; plan.factor_flops += r2 * r2; }
      }
      if (na >= ((int64_t)1 << 31)) { err = "too many assembly records"; return false; }
      plan.bnd_rows.reserve((size_t)nb + 64 * 1024); plan.child_map.reserve((size_t)nm + 64 * 1024); plan.asm_recs.reserve((size_t)na + 96 * 1024);
      hint_huge(plan.bnd_rows); hint_huge(plan.child_map); hint_huge(plan.asm_recs);
      plan.bnd_rows.resize((size_t)nb); plan.child_map.resize((size_t)nm); plan.asm_recs.resize((size_t)na); }
    parallel_chunks(S, 512, [&](int64_t b0, int64_t e0, int) {
        std::unique_ptr<int32_t[]> loc(new int32_t[(size_t)B.nv]);      // row of a vertex inside the current front (only entries set below are read: not initialised — 3.5 MB per thread for a shard of 8 x 100k poses)
        std::vector<AsmRec> recs, uniq, dup;
        for (int s = (int)b0; s < (int)e0; ++s) {
            Front &F = plan.fronts[s];
            { int32_t *o = &plan.bnd_rows[(size_t)F.bnd_off];
              for (int w : bndv[s]) for (int t = 0; t < B.dim(w); ++t) *o++ = gidx[w] + t; }
            int r = 0;
            for (int v : B.sn[s]) { loc[v] = r; r += B.dim(v); }
            for (int w : bndv[s]) { loc[w] = r; r += B.dim(w); }
            // extend-add maps of the children into this front
            for (int c : kids[s]) { int32_t *o = &plan.child_map[(size_t)plan.fronts[c].map_off];
                for (int w : bndv[c]) for (int t = 0; t < B.dim(w); ++t) *o++ = loc[w] + t; }
            // original entries
            recs.clear();
            if (!F.opaque) for (int v : B.sn[s]) {
                if (v < B.nfp) recs.push_back({ASM_POSE_DIAG, B.pose_of_fp[v], loc[v], loc[v]});
                else recs.push_back({ASM_LM_DIAG, B.lm_of_fl[v - B.nfp], loc[v], loc[v]});
                B.for_inc(v, [&](const Builder::Inc &e) {
                    if (vpos[e.other] <= vpos[v]) return;               // the earlier endpoint owns the block
                    int kind;
                    switch (e.kind) {                                    // e.kind describes v's role; `other` is the later vertex
                        case 0: kind = ASM_PP_T; break;                  // v = i earlier, j later: F = Hpp_off^T
                        case 1: kind = ASM_PP; break;                    // v = j earlier, i later
                        case 2: kind = ASM_PL_T; break;                  // v = pose earlier, landmark later
                        default: kind = ASM_PL; break;                   // v = landmark earlier, pose later
                    }
                    recs.push_back({kind, e.epos, loc[e.other], loc[v]});
                });
            }
            // duplicates (parallel edges between the same two vertices) go to the tail
            std::stable_sort(recs.begin(), recs.end(), [](const AsmRec &x, const AsmRec &y) {
                return x.r0 != y.r0 ? x.r0 < y.r0 : x.c0 < y.c0; });
            uniq.clear(); dup.clear();
            for (size_t t = 0; t < recs.size(); ++t) {
                if (t > 0 && recs[t].r0 == recs[t - 1].r0 && recs[t].c0 == recs[t - 1].c0) dup.push_back(recs[t]);
                else uniq.push_back(recs[t]);
            }
            F.asm_dup = (int32_t)dup.size();
            std::copy(uniq.begin(), uniq.end(), plan.asm_recs.begin() + F.asm_off);
            std::copy(dup.begin(), dup.end(), plan.asm_recs.begin() + F.asm_off + (int64_t)uniq.size());
        } });
    GS_PT(5);
    // ---- levels ----
    int nlev = 0;
    for (auto &F : plan.fronts) nlev = std::max(nlev, F.level + 1);
    plan.level_start.assign(nlev + 1, 0);
    for (auto &F : plan.fronts) plan.level_start[F.level + 1]++;
    for (int l = 0; l < nlev; ++l) plan.level_start[l + 1] += plan.level_start[l];
    plan.level_fronts.resize(S);
    { std::vector<int32_t> fill(plan.level_start.begin(), plan.level_start.end() - 1);
      for (int s = 0; s < S; ++s) plan.level_fronts[fill[plan.fronts[s].level]++] = s; }
    // fronts beyond a wave (f > 63) are launched by size class (5 / 7 / 10 tile rows, each with its own LDS need): keep the classes
    // of a level together.  Stable, so a plan without such fronts keeps its order.
    if (plan.max_front > 63) {
        auto cls = [&](int s) { const int f = plan.fronts[s].npiv + plan.fronts[s].nbnd; return f <= 63 ? 0 : (f <= 79 ? 1 : (f <= 111 ? 2 : 3)); };
        for (int l = 0; l < nlev; ++l)
            std::stable_sort(plan.level_fronts.begin() + plan.level_start[l], plan.level_fronts.begin() + plan.level_start[l + 1],
                             [&](int a, int b) { return cls(a) < cls(b); });
    }
    GS_PT(6);
    // ---- pose-window shards: rank r owns the subtrees whose poses all lie in the r-th contiguous
    // window of the free-pose sequence; every front above them is "shared" (owner -1): the window-boundary
    // separator poses and the landmarks seen from more than one window.  An edge is evaluated by exactly one rank,
    // one that knows both endpoint estimates (owner of an interior endpoint, else the window of the pose).
    plan.world = std::max(1, opt.world); plan.rank = opt.rank;
    plan.pl_rank.reserve((size_t)Epl + TAIL_PL); plan.pp_rank.reserve((size_t)Epp + TAIL_PP);
    hint_huge(plan.pl_rank);
    plan.pose_known.reserve((size_t)N + TAIL_POSES); plan.lm_known.reserve((size_t)M + TAIL_LMS);
    pfill(plan.pl_rank, (size_t)Epl, (int32_t)(B.by_window ? -1 : 0)); pfill(plan.pp_rank, (size_t)Epp, (int32_t)(B.by_window ? -1 : 0));      // (by windows: -1 = an edge of another window's interior, somebody else's)
    pfill(plan.pose_known, (size_t)N, (uint8_t)1); pfill(plan.lm_known, (size_t)M, (uint8_t)1);
    plan.level_start_owned = plan.level_start; plan.level_fronts_owned = plan.level_fronts;
    plan.level_start_shared.assign(nlev + 1, 0);
    if (plan.world > 1) {
        const int W = plan.world;
        auto window = [&](int fpos) { return (int)((int64_t)fpos * W / std::max(1, B.nfp)); };
        std::vector<int32_t> wmin(S, INT32_MAX), wmax(S, -1);
        parallel_chunks(S, 512, [&](int64_t s0, int64_t s1, int) { for (int s = (int)s0; s < (int)s1; ++s) {      // a supernode's own poses ...
            if (B.by_window && is_opaque(s)) wmin[s] = wmax[s] = B.opaque_of_pose[B.sn[s][0]];      // (another rank's window: no walk over its ~100k poses)
            else for (int v : B.sn[s]) if (v < B.nfp) {
                const int w = B.by_window ? (int)(std::upper_bound(B.wf.begin(), B.wf.end(), v) - B.wf.begin()) - 1 : window(v);      // (the table of window starts instead of a 64-bit division per pose)
                wmin[s] = std::min(wmin[s], w); wmax[s] = std::max(wmax[s], w); } } });
        for (int s = 0; s < S; ++s)                                   // ... and its children's, bottom-up
            for (int c : kids[s]) { wmin[s] = std::min(wmin[s], wmin[c]); wmax[s] = std::max(wmax[s], wmax[c]); }
        for (int s = S - 1; s >= 0; --s) {
            Front &F = plan.fronts[s];
            if (wmax[s] < 0) F.owner = parent[s] >= 0 ? plan.fronts[parent[s]].owner : -1;      // no pose below: follow the parent
            else F.owner = (wmin[s] == wmax[s]) ? wmin[s] : -1;
        }
        auto vowner = [&](int v) { return plan.fronts[sn_of[v]].owner; };
        auto rank_of_pl = [&](int64_t k) { const int fp = B.fp_of_pose[g.pl_p[(size_t)k]], fl = B.fl_of_lm[g.pl_l[(size_t)k]];
            int r = 0;
            if (fp >= 0 && vowner(fp) >= 0) r = vowner(fp);
            else if (fl >= 0 && vowner(B.nfp + fl) >= 0) r = vowner(B.nfp + fl);
            else if (fp >= 0) r = window(fp);
            return r; };
        auto rank_of_pp = [&](int64_t k) { const int fi = B.fp_of_pose[g.pp_i[(size_t)k]], fj = B.fp_of_pose[g.pp_j[(size_t)k]];
            int r = 0;
            if (fi >= 0 && vowner(fi) >= 0) r = vowner(fi);
            else if (fj >= 0 && vowner(fj) >= 0) r = vowner(fj);
            else if (fi >= 0) r = window(fi);
            else if (fj >= 0) r = window(fj);
            return r; };
        if (B.by_window) {                                            // the edges of the poses this rank's plan is built from: every edge it evaluates is among them
            parallel_chunks((int64_t)ing.size(), 4096, [&](int64_t b0, int64_t e0, int) { for (int64_t u = b0; u < e0; ++u) { const int p = ing[(size_t)u];
                for (int q = plan.pl_start[p]; q < plan.pl_start[p + 1]; ++q) plan.pl_rank[(size_t)q] = rank_of_pl(q);
                for (int q = plan.ppadj_start[p]; q < plan.ppadj_start[p + 1]; ++q) { const int code = plan.ppadj[q], k = code >> 1;
                    if (!(code & 1) || !ing_flag[g.pp_i[k]]) plan.pp_rank[(size_t)k] = rank_of_pp(k); } } });      // (an edge between two such poses: written from its first one — one writer per entry)
        } else {
        parallel_chunks(Epl, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t k = b0; k < e0; ++k) plan.pl_rank[(size_t)k] = rank_of_pl(k); });
        parallel_chunks(Epp, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t k = b0; k < e0; ++k) plan.pp_rank[(size_t)k] = rank_of_pp(k); });
        }
        parallel_chunks(B.nfp, 16384, [&](int64_t b0, int64_t e0, int) { for (int i = (int)b0; i < (int)e0; ++i) { const int o = vowner(i); plan.pose_known[B.pose_of_fp[i]] = (o < 0 || o == plan.rank); } });
        parallel_chunks(B.nfl, 16384, [&](int64_t b0, int64_t e0, int) { for (int l = (int)b0; l < (int)e0; ++l) { const int o = vowner(B.nfp + l); plan.lm_known[B.lm_of_fl[l]] = (o < 0 || o == plan.rank); } });
        // per-rank level lists: owned fronts, then the shared top; exchange slots of the shared fronts
        plan.level_start_owned.assign(nlev + 1, 0); plan.level_fronts_owned.clear(); plan.level_fronts_shared.clear();
        for (int l = 0; l < nlev; ++l) {
            for (int q = plan.level_start[l]; q < plan.level_start[l + 1]; ++q) { const int s = plan.level_fronts[q];
                if (plan.fronts[s].owner == plan.rank) plan.level_fronts_owned.push_back(s);
                else if (plan.fronts[s].owner < 0) plan.level_fronts_shared.push_back(s); }
            plan.level_start_owned[l + 1] = (int32_t)plan.level_fronts_owned.size();
            plan.level_start_shared[l + 1] = (int32_t)plan.level_fronts_shared.size();
        }
        plan.n_shared_fronts = (int32_t)plan.level_fronts_shared.size();
        // exchange slots in ELIMINATION order of the shared fronts: the one order every rank agrees on (levels differ between
        // ranks — a rank sees the other ranks' subtrees as single supernodes)
        plan.x_off.assign(S, -1);
        for (int s = 0; s < S; ++s) if (plan.fronts[s].owner < 0) { const Front &F = plan.fronts[s]; const int64_t f = F.npiv + F.nbnd;
            plan.x_off[s] = plan.exchange_doubles; plan.exchange_doubles += (f + 1) * f; }
        plan.exchange_doubles += 2;                // tail: [0] the ranks' failure flags (summed by the same all-reduce), [1] spare
    }
    else if (opt.force_shared_top > 0 && nlev > 1) {
        // gs_debug_options.force_shared_top (world 1 only): the top k levels are treated like the shared top of a sharded graph — this
        // rank's contribution goes to the exchange buffer, the caller (or gs_dist_iterate: RCCL) all-reduces it over a group of one, the
        // top is then factorised from the buffer — so that the whole collective path runs with a NON-EMPTY exchange buffer on one GPU.
        // Upward closed: a parent's level exceeds its children's.
        const int k = std::min(opt.force_shared_top, nlev - 1);
        for (int s = 0; s < S; ++s) plan.fronts[s].owner = plan.fronts[s].level >= nlev - k ? -1 : 0;
        plan.level_start_owned.assign(nlev + 1, 0); plan.level_fronts_owned.clear(); plan.level_fronts_shared.clear();
        for (int l = 0; l < nlev; ++l) {
            for (int q = plan.level_start[l]; q < plan.level_start[l + 1]; ++q) { const int s = plan.level_fronts[q];
                if (plan.fronts[s].owner < 0) plan.level_fronts_shared.push_back(s); else plan.level_fronts_owned.push_back(s); }
            plan.level_start_owned[l + 1] = (int32_t)plan.level_fronts_owned.size();
            plan.level_start_shared[l + 1] = (int32_t)plan.level_fronts_shared.size(); }
        plan.n_shared_fronts = (int32_t)plan.level_fronts_shared.size();
        plan.x_off.assign(S, -1);
        for (int s = 0; s < S; ++s) if (plan.fronts[s].owner < 0) { const Front &F = plan.fronts[s]; const int64_t f = F.npiv + F.nbnd;
            plan.x_off[s] = plan.exchange_doubles; plan.exchange_doubles += (f + 1) * f; }
        plan.exchange_doubles += 2;
    }
    plan.dist = plan.world > 1 || plan.n_shared_fronts > 0;
    // ---- the bottom subtrees (k_factor3_sub: a level-1 front and the leaves below it in one workgroup): this rank's level-0 fronts in two
    // runs — first the leaves whose parent sits higher up (or belongs to somebody else): they keep the leaf launch —, then the leaves
    // under this rank's level-1 fronts, which get no launch of their own.  Stable, plans without wave-only fronts keep their class order.
    if (plan.max_front <= 63 && nlev >= 2) {
        auto under_l1 = [&](int s) { const int P = plan.fronts[s].parent; return P >= 0 && plan.fronts[P].level == 1 && plan.fronts[P].owner == plan.fronts[s].owner && !plan.fronts[P].opaque; };
        std::stable_partition(plan.level_fronts_owned.begin() + plan.level_start_owned[0], plan.level_fronts_owned.begin() + plan.level_start_owned[1],
                              [&](int s) { return !under_l1(s); });
        if (!plan.dist) plan.level_fronts = plan.level_fronts_owned;
    }
    // ---- device layout of the observation edges: ELL, T lanes per pose, over the poses THIS RANK SWEEPS ------------------
    // (after the shard assignment: a rank lays out, uploads and linearises the pose range [ell_p0, ell_p0 + ell_np) that holds its
    // edges — its window plus the boundary poses of the shared top — and nothing of the other windows: 1 / world of the edge
    // streams, of the H_pl blocks and of this phase.  World 1: every pose.)
    // The s-th edge of pose p sits at  idx = (s / T) * (T * np) + T * (p - p0) + (s % T):  the T lanes of a pose each own up to
    // R = ceil(Kmax / T) edges (slots i = 0 .. R-1), consecutive lanes read consecutive addresses in every slot, and the loads of
    // all R slots of a thread are independent (memory-level parallelism instead of occupancy).  T is the smallest power of two
    // with R <= 4; graphs with a pose of more than 32 observations use T = 8 and a larger R, which only the gather kernels
    // handle.  The LAST entry of the layout is a permanent empty slot: the assembly record of an edge another rank evaluates
    // points there (its block reads as zero).
    int plo = 0, phi = N;
    if (plan.world > 1) { plo = N; phi = 0;
        std::vector<int> lo_t(host_threads() + 1, N), hi_t(host_threads() + 1, 0);
        parallel_chunks(N, 16384, [&](int64_t b0, int64_t e0, int t) { int lo2 = N, hi2 = 0;
            for (int p = (int)b0; p < (int)e0; ++p) { bool any = false;
                if (B.by_window && !ing_flag[p]) continue;
                for (int s2 = plan.pl_start[p]; s2 < plan.pl_start[p + 1] && !any; ++s2) any = plan.pl_rank[plan.pl_order[s2]] == plan.rank;
                for (int q = plan.ppadj_start[p]; q < plan.ppadj_start[p + 1] && !any; ++q) any = plan.pp_rank[plan.ppadj[q] >> 1] == plan.rank;
                if (any) { lo2 = std::min(lo2, p); hi2 = std::max(hi2, p + 1); } }
            lo_t[t] = lo2; hi_t[t] = hi2; });
        for (size_t t = 0; t < lo_t.size(); ++t) { plo = std::min(plo, lo_t[t]); phi = std::max(phi, hi_t[t]); }
        if (phi <= plo) { plo = 0; phi = 0; } }
GS_PT(70);
    const int kmax = kmax_all;
    int T = 1;
    while (T < 8 && (kmax + T - 1) / T > LIN_R) T *= 2;
    // fewer, fatter lanes give more loads in flight per wave, but the chip wants >= ~3 waves per SIMD
    // (1024 SIMDs): small graphs take more lanes per pose (measured on MI355X: 100k poses T=2 30 us vs T=4 40 us;
    // 10k poses T=2 13.6 us vs T=4 8.5 us)
    while (T < 8 && (int64_t)std::max(phi - plo, 1) * T / 64 < 3072) T *= 2;
    if (opt.ell_lanes == 1 || opt.ell_lanes == 2 || opt.ell_lanes == 4 || opt.ell_lanes == 8) T = opt.ell_lanes;   // tuning override (too few lanes => gather kernels)
    const int R = std::max(1, (kmax + T - 1) / T);
    const int PW = 64 / T, WT = (N + PW - 1) / PW;
    plan.wt_lo = plo / PW; plan.wt_hi = (phi + PW - 1) / PW;        // whole wave tiles
    const int p0 = plan.wt_lo * PW, p1 = std::min(N, plan.wt_hi * PW), np_ = std::max(p1 - p0, 0);
    plan.ell_T = T; plan.ell_R = R; plan.ell_p0 = p0; plan.ell_np = np_;
    plan.ell_len = (int64_t)R * T * np_ + 1;
    plan.lin_ell_ok = R <= LIN_R;
    plan.ell_ins.reserve((size_t)plan.ell_len + TAIL_PL); plan.ell_of_ins.reserve((size_t)Epl + TAIL_PL);
    hint_huge(plan.ell_ins); hint_huge(plan.ell_of_ins);
    plan.ell_ins.assign((size_t)plan.ell_len, -1);
    pfill(plan.ell_of_ins, (size_t)Epl, (int32_t)-1);
    parallel_chunks(np_, 8192, [&](int64_t b0, int64_t e0, int) {
        for (int p = p0 + (int)b0; p < p0 + (int)e0; ++p)
            for (int s2 = 0; s2 < plan.pl_start[p + 1] - plan.pl_start[p]; ++s2) {
                const int64_t idx = (int64_t)(s2 / T) * ((int64_t)T * np_) + (int64_t)T * (p - p0) + (s2 % T);
                const int k = plan.pl_order[plan.pl_start[p] + s2];
                plan.ell_ins[(size_t)idx] = k; plan.ell_of_ins[k] = (int32_t)idx; } });
GS_PT(71);
    // the assembly records named observation edges by insertion index so far
    { const int32_t zero_slot = (int32_t)(plan.ell_len - 1);
      parallel_chunks((int64_t)plan.asm_recs.size(), 16384, [&](int64_t b0, int64_t e0, int) {
          for (int64_t t = b0; t < e0; ++t) { AsmRec &r = plan.asm_recs[(size_t)t];
              if (r.kind == ASM_PL || r.kind == ASM_PL_T) { const int32_t e = plan.ell_of_ins[r.src]; r.src = e >= 0 ? e : zero_slot; } } }); }
GS_PT(72);
    // landmark -> ELL indices of its edges (the gather kernels: single GPU only)
    plan.lm_edges.clear();
    if (plan.world == 1) { plan.lm_edges.resize(Epl);
        parallel_chunks(Epl, 16384, [&](int64_t b0, int64_t e0, int) { for (int64_t q = b0; q < e0; ++q) plan.lm_edges[(size_t)q] = plan.ell_of_ins[lm_k[(size_t)q]]; }); }
    else plan.lm_edges.assign(1, 0);
GS_PT(73);
    // ---- wave tiles of the fused A5-A7 kernel: one wave = 64/T consecutive poses.  Per wave tile the distinct
    // landmarks it touches ("groups") with the wave-local positions (slot*64 + lane) of their edges.  Partial-sum
    // slots are ordered by (landmark, wave tile): the finalize pass reads one contiguous run per landmark and the
    // summation order is fixed => bitwise reproducible without atomics.  Only the tiles [wt_lo, wt_hi) have groups.
    if (plan.lin_ell_ok) {
        plan.n_wtiles = WT;
        plan.wt_grp_start.assign(WT + 1, 0);
        big_assign(plan.ell_dst, (size_t)plan.ell_len, (uint16_t)0xFFFF);
        // every wave tile is independent: chunks of tiles on the host threads, each into its own lists, stitched afterwards
        const int C = chunk_count(std::max(1, plan.wt_hi - plan.wt_lo), 256);
        struct TileOut { std::vector<int32_t> grp_lm, grp_pos_start, grp_pos, tile_groups; };
        std::vector<TileOut> outs(C);
        std::vector<int64_t> cb(C + 1, 0);
        for (int c = 0; c <= C; ++c) cb[c] = plan.wt_lo + (int64_t)(plan.wt_hi - plan.wt_lo) * c / C;
        parallel_chunks(C, 1, [&](int64_t c0, int64_t c1, int) {
            for (int c = (int)c0; c < (int)c1; ++c) { TileOut &O = outs[c];
                { const size_t nt = (size_t)(cb[c + 1] - cb[c]);     // (no re-allocation while the chunk's lists grow)
                  O.grp_pos.reserve(nt * 64 * (size_t)R); O.grp_lm.reserve(nt * 48); O.grp_pos_start.reserve(nt * 48); O.tile_groups.reserve(nt); }
                std::vector<std::pair<int32_t, int32_t>> tmp;          // (landmark, local position)
                tmp.reserve(64 * (size_t)R);
                for (int w = (int)cb[c]; w < (int)cb[c + 1]; ++w) {
                    tmp.clear();
                    for (int lane = 0; lane < 64; ++lane) { const int p = w * PW + lane / T, h = lane % T;
                        if (p >= N) break;
                        for (int i = 0; i < R; ++i) { const int s2 = i * T + h;
                            if (s2 < plan.pl_start[p + 1] - plan.pl_start[p]) tmp.emplace_back(g.pl_l[plan.pl_order[plan.pl_start[p] + s2]], i * 64 + lane); } }
                    std::sort(tmp.begin(), tmp.end());
                    int ng = 0;
                    for (size_t i = 0; i < tmp.size(); ++i) {
                        if (i == 0 || tmp[i].first != tmp[i - 1].first) { O.grp_lm.push_back(tmp[i].first); O.grp_pos_start.push_back((int32_t)O.grp_pos.size()); ++ng; }
                        O.grp_pos.push_back(tmp[i].second);
                        plan.ell_dst[(size_t)((int64_t)(tmp[i].second >> 6) * ((int64_t)T * np_) + (int64_t)(w - plan.wt_lo) * 64 + (tmp[i].second & 63))] = (uint16_t)i;
                    }
                    O.tile_groups.push_back(ng);
                } } });
GS_PT(74);
        { size_t ng = 0, np2 = 0; for (auto &O : outs) { ng += O.grp_lm.size(); np2 += O.grp_pos.size(); }
          plan.grp_lm.reserve(ng); plan.grp_pos_start.reserve(ng + 1); plan.grp_pos.reserve(np2);
          int w = plan.wt_lo;                                          // tiles below wt_lo have no groups: wt_grp_start stays 0 there
          for (auto &O : outs) { const int32_t pos0 = (int32_t)plan.grp_pos.size();
              plan.grp_lm.insert(plan.grp_lm.end(), O.grp_lm.begin(), O.grp_lm.end());
              for (int32_t v : O.grp_pos_start) plan.grp_pos_start.push_back(pos0 + v);
              plan.grp_pos.insert(plan.grp_pos.end(), O.grp_pos.begin(), O.grp_pos.end());
              for (int n : O.tile_groups) { plan.wt_grp_start[w + 1] = plan.wt_grp_start[w] + n; ++w; } }
          for (; w < WT; ++w) plan.wt_grp_start[w + 1] = plan.wt_grp_start[w]; }
GS_PT(75);
        plan.grp_pos_start.push_back((int32_t)plan.grp_pos.size());
        plan.wt_desc.resize((size_t)WT * 4);
        for (int w = 0; w < WT; ++w) { const int a = plan.wt_grp_start[w], b = plan.wt_grp_start[w + 1];
            plan.wt_desc[4 * (size_t)w] = a; plan.wt_desc[4 * (size_t)w + 1] = b - a;
            plan.wt_desc[4 * (size_t)w + 2] = plan.grp_pos_start[a]; plan.wt_desc[4 * (size_t)w + 3] = plan.grp_pos_start[b] - plan.grp_pos_start[a]; }
        const int G = (int)plan.grp_lm.size();
        plan.lm_grp_start.assign(M + 1, 0);
        for (int q = 0; q < G; ++q) plan.lm_grp_start[plan.grp_lm[q] + 1]++;
        for (int l = 0; l < M; ++l) plan.lm_grp_start[l + 1] += plan.lm_grp_start[l];
        plan.grp_slot.resize(G);
        std::vector<int32_t> fill(plan.lm_grp_start.begin(), plan.lm_grp_start.end() - 1);
        for (int q = 0; q < G; ++q) plan.grp_slot[q] = fill[plan.grp_lm[q]]++;
    }

GS_PT(76);
    plan.base_N = plan.planned_N = N; plan.base_M = plan.planned_M = M; plan.base_Epp = plan.planned_Epp = Epp; plan.base_Epl = plan.planned_Epl = Epl;
    plan.n_growths = 0; plan.reshape_version = g.reshape_version; plan.front_limit = plan.max_front > 63 ? 159 : 63;
    plan.root_f0 = plan.fronts.empty() ? 0 : plan.fronts.back().npiv + plan.fronts.back().nbnd;
    // room for grow_plan's re-written runs: without it the first growth step pays for reallocating (and copying) these arrays — 3 of
    // its 4 ms at 100k poses
    plan.bnd_rows.reserve(plan.bnd_rows.size() + 64 * 1024); plan.child_map.reserve(plan.child_map.size() + 64 * 1024);
    plan.asm_recs.reserve(plan.asm_recs.size() + 96 * 1024);
    plan.pose_gidx.reserve((size_t)N + TAIL_POSES); plan.pose_known.reserve((size_t)N + TAIL_POSES);
    plan.lm_gidx.reserve((size_t)M + TAIL_LMS); plan.lm_known.reserve((size_t)M + TAIL_LMS);
    plan.pl_order.reserve((size_t)Epl + TAIL_PL); plan.ell_of_ins.reserve((size_t)Epl + TAIL_PL); plan.pl_rank.reserve((size_t)Epl + TAIL_PL);
    plan.ell_ins.reserve((size_t)plan.ell_len + TAIL_PL); plan.pp_order.reserve((size_t)Epp + TAIL_PP); plan.pp_rank.reserve((size_t)Epp + TAIL_PP);
    plan.valid = true;
    GS_PT(7);

    plan.ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// ---- append-only growth -------------------------------------------------------------------------------------------------
// New poses (and new landmarks: they can only touch new poses) are eliminated LAST: extra pivots of the root front.  A neighbour v of a new pose P (the other end of one of P's
// edges, free, older) is a pivot of some front s0; eliminating v makes P a neighbour of everything v touched later, i.e. P joins
// the boundary of s0 and of every front on the way from s0 to the root — the exact fill of the enlarged matrix under the enlarged
// order.  P's scalars get the largest elimination indices, so they are the LAST rows of every front they enter: no existing row of
// any front moves, no existing child map entry changes; runs that grow (boundary rows, child map, assembly records of a front)
// are re-written at the end of their arrays.  The block of an edge (v, P) lands in s0 (v is the earlier end), P's diagonal block in
// the root.  Edge sources beyond the base counts name the tail arenas of the device: observation edge k -> virtual layout index
// ell_len + (k - base_Epl), odometry edge k -> k, pose p -> p.
bool grow_plan(const HostGraph &g, Plan &P, Growth &out, std::string &why) {
    out = Growth();
    auto no = [&](const char *m) { why = m; return false; };
    if (!P.valid) return no("no plan");
    if (P.dist) return no("sharded plan");
    if (!P.lin_ell_ok || P.max_front > 159) return no("plan outside the matrix-core forms");
    if (g.reshape_version != P.reshape_version) return no("a fixed flag changed (or the graph was cleared)");
    const int N0 = P.planned_N, N1 = g.n_poses(), M0 = P.planned_M, M1 = g.n_lms(), Epp0 = P.planned_Epp, Epp1 = g.n_pp(), Epl0 = P.planned_Epl, Epl1 = g.n_pl();
    if (N1 < N0 || M1 < M0 || Epp1 < Epp0 || Epl1 < Epl0) return no("graph shrank");
    if (N1 == N0 && M1 == M0 && Epp1 == Epp0 && Epl1 == Epl0) return no("nothing new");
    if (N1 - P.base_N > TAIL_POSES || M1 - P.base_M > TAIL_LMS || Epl1 - P.base_Epl > TAIL_PL || Epp1 - P.base_Epp > TAIL_PP) return no("tail capacity");
    const int S = (int)P.fronts.size();
    if (S == 0) return no("empty plan");
    const int R = S - 1;
    for (int s = 0; s < S; ++s) if ((P.fronts[s].parent < 0) != (s == R)) return no("forest: more than one root");
    if (P.fronts[R].nbnd != 0) return no("root with a boundary");
    // every new edge must have a NEW pose at the pose end (observation) / at one end (odometry): an old pose's observation edges
    // sit in the linearisation layout, which does not grow
    for (int k = Epl0; k < Epl1; ++k) if (g.pl_p[k] < N0) return no("new observation edge on an old pose");
    for (int k = Epp0; k < Epp1; ++k) if (g.pp_i[k] < N0 && g.pp_j[k] < N0) return no("new odometry edge between old poses");
    for (int p = N0; p < N1; ++p) if (g.pose_fixed[p]) return no("new pose is fixed");
    for (int l = M0; l < M1; ++l) if (g.lm_fixed[l]) return no("new landmark is fixed");
    // front of a scalar: fronts are in elimination order with contiguous pivots (anything beyond the old scalars: the root)
    std::vector<int32_t> piv0(S);
    for (int s = 0; s < S; ++s) piv0[s] = P.fronts[s].piv0;
    auto front_of = [&](int gi) { return (int)(std::upper_bound(piv0.begin(), piv0.end(), gi) - piv0.begin()) - 1; };
    // the batch's vertices enter the order as: new poses (by index), then new landmarks (by index)
    const int nP = N1 - N0, nL = M1 - M0, nV = nP + nL;
    auto vdim = [&](int v) { return v < nP ? 3 : 2; };
    std::vector<int32_t> gnew(nV);
    { int sc = P.n_scalar; for (int v = 0; v < nV; ++v) { gnew[v] = sc; sc += vdim(v); } }
    auto pose_g = [&](int p) { return p < N0 ? P.pose_gidx[p] : gnew[p - N0]; };
    auto lm_g = [&](int l) { return l < M0 ? P.lm_gidx[l] : gnew[nP + (l - M0)]; };
    // neighbours: for every new vertex the OLDER free end of each of its new edges (the block of an edge lands in the front of its
    // earlier end, at the later end's rows)
    struct Nb { int32_t gv, kind, src; };                    // earlier end's first scalar, record kind, record source
    std::vector<std::vector<Nb>> nbs(nV);
    std::vector<uint8_t> lm_seen(nL, 0);
    for (int k = Epl0; k < Epl1; ++k) { const int p = g.pl_p[k], l = g.pl_l[k];
        if (g.lm_fixed[l]) continue;                         // fixed landmark: the edge only feeds the pose's diagonal block
        const int gp = pose_g(p), gl = lm_g(l);
        if (gl < 0) return no("free landmark without a scalar");
        if (l < M0 && l < P.base_M && P.lm_grp_start[l + 1] <= P.lm_grp_start[l]) return no("landmark without a partial-sum slot");
        if (l >= M0) lm_seen[l - M0] = 1;
        const int src = (int32_t)(P.ell_len + (k - P.base_Epl));
        if (gl < gp) { auto &v = nbs[p - N0];                // landmark earlier: rows of the pose below the landmark's columns
            for (const Nb &o : v) if (o.kind == ASM_PL && o.gv == gl) return no("a new pose observes a landmark twice");
            v.push_back({gl, ASM_PL, src}); }
        else { auto &v = nbs[nP + (l - M0)];                 // (a new landmark seen by a pose of the same batch) pose earlier
            for (const Nb &o : v) if (o.kind == ASM_PL_T && o.gv == gp) return no("a new pose observes a landmark twice");
            v.push_back({gp, ASM_PL_T, src}); } }
    for (int l = M0; l < M1; ++l) if (!lm_seen[l - M0]) return no("new landmark without an observation");
    for (int k = Epp0; k < Epp1; ++k) { const int i = g.pp_i[k], j = g.pp_j[k];
        if (g.pose_fixed[i] || g.pose_fixed[j]) continue;    // (a fixed end: diagonal contribution only)
        const int later = std::max(i, j), earlier = std::min(i, j);      // poses enter the order by index: the larger index is eliminated later
        if (later < N0) continue;
        const int ge = pose_g(earlier);
        if (ge < 0) continue;
        for (const Nb &o : nbs[later - N0]) if ((o.kind == ASM_PP || o.kind == ASM_PP_T) && o.gv == ge) return no("parallel odometry edges on a new pose");
        nbs[later - N0].push_back({ge, later == j ? ASM_PP_T : ASM_PP, k}); }   // earlier end = i: F = Hpp_off^T (rows of j below i's columns)
    // ---- size check before anything is written
    { std::vector<int32_t> add(S, 0), stamp(S, -1);
      for (int v = 0; v < nV; ++v) { add[R] += vdim(v);
          for (const Nb &o : nbs[v]) { const int s0 = front_of(o.gv);
              for (int s = s0; s != R; s = P.fronts[s].parent) { if (stamp[s] == v) break; stamp[s] = v; add[s] += vdim(v); } } }
      // in a plan that holds workgroup fronts a front may grow to 159 scalars (a wave front of the path may become a workgroup front).
      // The ROOT, which every new vertex enters, at most 24 scalars beyond what the full phase gave it: grown without bound it costs
      // every iteration more than the structure phases it saves (lap-sized graphs: 2.5 -> 4.0 ms per optimize(10) over 16 keyframes,
      // scripts/keyframe_stream.py)
      for (int s = 0; s < S; ++s) { if (!add[s]) continue;
          const int f0 = P.fronts[s].npiv + P.fronts[s].nbnd, lim = s == R ? std::min(P.front_limit, std::max(63, P.root_f0 + 24)) : P.front_limit;
          if (f0 + add[s] > lim) return no(lim == 63 ? "a front would exceed 63 scalars" : "a front would exceed 159 scalars"); } }
    // ---- apply
    out.bnd_from = (int64_t)P.bnd_rows.size(); out.map_from = (int64_t)P.child_map.size(); out.asm_from = (int64_t)P.asm_recs.size();
    out.first_pose = N0; out.first_lm = M0; out.first_pp = Epp0; out.first_pl = Epl0;
    std::vector<uint8_t> touched(S, 0);
    std::vector<int32_t> stamp(S, -1), path;
    std::vector<std::vector<AsmRec>> newrec(S);
    for (int v = 0; v < nV; ++v) {
        Front &Rf = P.fronts[R];
        const int dv = vdim(v), gV = gnew[v], rowR = Rf.npiv;
        Rf.npiv += dv; touched[R] = 1;
        if (v < nP) P.pose_gidx.push_back(gV); else P.lm_gidx.push_back(gV);
        path.clear();
        for (const Nb &o : nbs[v]) { const int s0 = front_of(o.gv);
            for (int s = s0; s != R; s = P.fronts[s].parent) { if (stamp[s] == v) break; stamp[s] = v; path.push_back(s); } }
        std::sort(path.begin(), path.end());
        for (int s : path) { P.fronts[s].nbnd += dv; touched[s] = 1; }
        for (int s : path) { Front &F = P.fronts[s]; const int nb_old = F.nbnd - dv;
            // boundary rows: the old run + the new vertex's scalars, at the end of the array
            { const int64_t o = (int64_t)P.bnd_rows.size(); P.bnd_rows.resize((size_t)o + F.nbnd);
              std::copy(P.bnd_rows.begin() + F.bnd_off, P.bnd_rows.begin() + F.bnd_off + nb_old, P.bnd_rows.begin() + o);
              for (int t = 0; t < dv; ++t) P.bnd_rows[(size_t)o + nb_old + t] = gV + t;
              F.bnd_off = o; }
            // rows of the parent's front: the new vertex's are its last (the root: its newest pivots)
            { const Front &Pa = P.fronts[F.parent]; const int prow = F.parent == R ? rowR : Pa.npiv + Pa.nbnd - dv;
              const int64_t o = (int64_t)P.child_map.size(); P.child_map.resize((size_t)o + F.nbnd);
              std::copy(P.child_map.begin() + F.map_off, P.child_map.begin() + F.map_off + nb_old, P.child_map.begin() + o);
              for (int t = 0; t < dv; ++t) P.child_map[(size_t)o + nb_old + t] = prow + t;
              F.map_off = o; } }
        for (const Nb &o : nbs[v]) { const int s0 = front_of(o.gv); const Front &F = P.fronts[s0];
            const int r0 = s0 == R ? rowR : F.npiv + F.nbnd - dv;
            newrec[s0].push_back({o.kind, o.src, r0, o.gv - F.piv0}); }
        if (v < nP) newrec[R].push_back({ASM_POSE_DIAG, N0 + v, rowR, rowR});
        else newrec[R].push_back({ASM_LM_DIAG_TAIL, M0 + (v - nP), rowR, rowR});
    }
    P.n_scalar = gnew.empty() ? P.n_scalar : gnew.back() + vdim(nV - 1);
    for (int s = 0; s < S; ++s) { if (!touched[s]) continue;
        Front &F = P.fronts[s]; out.fronts.push_back(s);
        if (!newrec[s].empty()) {                                // unique records (the new ones have the largest rows), then the duplicates
            const int nu = F.asm_cnt - F.asm_dup; const int64_t o = (int64_t)P.asm_recs.size();
            P.asm_recs.resize((size_t)o + F.asm_cnt + newrec[s].size());
            std::copy(P.asm_recs.begin() + F.asm_off, P.asm_recs.begin() + F.asm_off + nu, P.asm_recs.begin() + o);
            std::copy(newrec[s].begin(), newrec[s].end(), P.asm_recs.begin() + o + nu);
            std::copy(P.asm_recs.begin() + F.asm_off + nu, P.asm_recs.begin() + F.asm_off + F.asm_cnt, P.asm_recs.begin() + o + nu + (int64_t)newrec[s].size());
            F.asm_off = (int32_t)o; F.asm_cnt += (int32_t)newrec[s].size(); }
        F.L_off = P.l_doubles; F.U_off = P.u_doubles;
        P.l_doubles += (int64_t)(F.npiv + F.nbnd + 1) * F.npiv; P.u_doubles += (int64_t)(F.nbnd + 1) * F.nbnd;
        P.max_front = std::max(P.max_front, F.npiv + F.nbnd); }
    for (int k = Epl0; k < Epl1; ++k) { const int32_t e = (int32_t)(P.ell_len + (k - P.base_Epl));
        P.pl_order.push_back(k); P.ell_of_ins.push_back(e);
        if ((int64_t)P.ell_ins.size() < (int64_t)e + 1) P.ell_ins.resize((size_t)e + 1, -1);
        P.ell_ins[(size_t)e] = k; }
    for (int k = Epp0; k < Epp1; ++k) P.pp_order.push_back(k);
    P.pose_known.resize(N1, 1); P.lm_known.resize(M1, 1); P.pl_rank.resize(Epl1, 0); P.pp_rank.resize(Epp1, 0);
    P.planned_N = N1; P.planned_M = M1; P.planned_Epp = Epp1; P.planned_Epl = Epl1; ++P.n_growths;
    return true;
}

// Flat dump: header[16] then arrays, all int32:
//   header: magic, n_scalar, n_fronts, n_levels, max_front, n_poses, n_lms, n_pl, n_pp, n_asm,
//           len(bnd_rows), len(child_map), len(children), ell_len, ell_T, ell_R
//   pose_gidx[n_poses] lm_gidx[n_lms] pl_order[n_pl] pp_order[n_pp] ell_ins[ell_len]
//   fronts[n_fronts][13]: npiv nbnd piv0 parent level owner bnd_off map_off asm_off asm_cnt asm_dup child_off child_cnt
//   bnd_rows child_map children asm_recs[n_asm][4] level_start[n_levels+1] level_fronts[n_fronts]
void export_plan(const Plan &p, std::vector<int32_t> &out) {
    out.clear();
    const int nlev = (int)p.level_start.size() - 1;
    int32_t hdr[16] = {0x47535031, p.n_scalar, (int32_t)p.fronts.size(), nlev, p.max_front,
                       (int32_t)p.pose_gidx.size(), (int32_t)p.lm_gidx.size(), (int32_t)p.pl_order.size(),
                       (int32_t)p.pp_order.size(), (int32_t)p.asm_recs.size(), (int32_t)p.bnd_rows.size(),
                       (int32_t)p.child_map.size(), (int32_t)p.children.size(), (int32_t)p.ell_ins.size() /* = ell_len; after grow_plan: + the tail's virtual indices */, p.ell_T, p.ell_R};
    out.insert(out.end(), hdr, hdr + 16);
    auto app = [&](const auto &v) { out.insert(out.end(), v.begin(), v.end()); };
    app(p.pose_gidx); app(p.lm_gidx); app(p.pl_order); app(p.pp_order); app(p.ell_ins);
    for (const auto &F : p.fronts) {
        int32_t r[13] = {F.npiv, F.nbnd, F.piv0, F.parent, F.level, F.owner, (int32_t)F.bnd_off, (int32_t)F.map_off,
                         F.asm_off, F.asm_cnt, F.asm_dup, F.child_off, F.child_cnt};
        out.insert(out.end(), r, r + 13);
    }
    app(p.bnd_rows); app(p.child_map); app(p.children);
    for (const auto &a : p.asm_recs) { out.push_back(a.kind); out.push_back(a.src); out.push_back(a.r0); out.push_back(a.c0); }
    app(p.level_start); app(p.level_fronts);
    // pose-window shards: world, rank, #shared fronts, exchange doubles, then pl_rank[n_pl] pp_rank[n_pp] (insertion
    // order), pose_known[n_poses], lm_known[n_lms], x_off[n_fronts] (only when world > 1)
    out.push_back(p.world); out.push_back(p.rank); out.push_back(p.n_shared_fronts); out.push_back((int32_t)p.exchange_doubles);
    if (p.world > 1) {
        app(p.pl_rank); app(p.pp_rank);
        for (auto v : p.pose_known) out.push_back(v);
        for (auto v : p.lm_known) out.push_back(v);
        for (auto v : p.x_off) out.push_back((int32_t)v);
    }
}

}  // namespace gs
