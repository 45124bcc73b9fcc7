// CholFactor: host-side construction of the panel/update programme, GPU numeric factorisation, and the
// cholsol solve phase (P, lsolve, ltsolve, P') on the device.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <climits>

#include "chol_engine.hpp"
#include "../host/parallel.hpp"

namespace rsp {

void CholFactor::join_pattern_jobs()
{
    for (auto &t : pat_thread_)
        if (t.joinable()) t.join();
}

int CholFactor::analyze(const CscView &A, const Symbolic &S, bool single_use)
{
    join_pattern_jobs();
    pat_lower_.reset();
    pat_upper_.reset();
    single_use_ = single_use;
    double tmark = now_s();
    auto lap = [&](const char *what) {
        if (getenv("RSP_TIMING")) fprintf(stderr, "chol analyze: %s %.3fs\n", what, now_s() - tmark);
        tmark = now_s();
    };
    n_ = A.n;
    annz_ = A.nnz();
    factor_valid_ = false;
    pattern_hash_ = pattern_hash(A);
    replicas_.clear();
    (void)hipGetDevice(&device_);
    has_pinv_ = S.has_pinv;
    pinv_ = S.pinv;
    if (n_ >= INT32_MAX) return kBadInput;
    if (has_pinv_) {  // must be a permutation
        std::vector<char> seen(n_, 0);
        for (i64 k = 0; k < n_; ++k) {
            if (pinv_[k] < 0 || pinv_[k] >= n_ || seen[pinv_[k]]) return kBadInput;
            seen[pinv_[k]] = 1;
        }
    }
    for (i64 j = 0; j < n_; ++j)
        if (S.parent[j] != -1 && (S.parent[j] <= j || S.parent[j] >= n_)) return kBadInput;

    // ---- pattern of L, exactly the reference's layout -------------------------------------------
    CscView Ap = A;
    Ap.x = nullptr;
    std::vector<i64> source;
    Csc C = symmetric_permute_upper(Ap, has_pinv_ ? pinv_.data() : nullptr, &source);
    if (!cholesky_pattern(C.view(), S.parent, S.cp, Li_)) return kBadInput;
    Lp_ = S.cp;
    lnz_ = Lp_[n_];
    flops_ = 0.0;
    for (i64 j = 0; j < n_; ++j) {
        const double c = (double)(Lp_[j + 1] - Lp_[j]);
        flops_ += c * c;
    }
    // A handle that will solve many times: its two solve programmes depend on the pattern alone as well, so they are
    // emitted beside this analysis and the factorisation too (placeholder values), and the factor's values reach them on
    // the device at the first solve (TriSolver::refresh_device) -- L never visits the host.
    const auto off = [](const char *name) { return getenv(name) && getenv(name)[0] == '0'; };
    const auto on = [](const char *name, char v) { return getenv(name) && getenv(name)[0] == v; };
    pat_ahead_ = want_programmes_ && !single_use && !off("RSP_REFILL") && !off("RSP_REFILL_DEVICE") && !off("RSP_PROGRAMME_AHEAD") && !on("RSP_TRI_SINGLE", '1') &&
                 !on("RSP_TRI_BLOCKED", '0');
    if ((single_use || pat_ahead_) && lnz_ < INT32_MAX) {
        // the solve phase needs the level structure of L in both directions, which depends on the pattern alone: start it
        // now, beside the rest of this analysis and the factorisation (Lp_ and Li_ do not change any more)
        pat_lower_.reset(new TriSolver);
        pat_upper_.reset(new TriSolver);
        int device = 0;
        RSP_HIP(hipGetDevice(&device));
        for (int k = 0; k < 2; ++k)
            pat_thread_[k] = std::thread([this, k, device] {
                try {
                    const CscView Lpat{n_, n_, Lp_.data(), Li_.data(), nullptr};
                    TriSolver &t = *(k == 0 ? pat_lower_ : pat_upper_);
                    if (pat_ahead_) t.keep_for_refill();
                    pat_status_[k] = t.analyze(Lpat, k);
                    if (!pat_ahead_ && single_use_ && pat_status_[k] == kOk && TriSolver::rows_wanted() && hipSetDevice(device) == hipSuccess)
                        t.prepare_rows_pattern();  // one vector, one use: the row kernel's pattern half goes to the device now
                    if (pat_ahead_ && pat_status_[k] == kOk) {
                        t.stage_blocked_pattern();
                        // ... and sends it to the device right away: the first solve finds it there
                        if (hipSetDevice(device) == hipSuccess && t.adopt_staged()) t.prepare_refresh_device(Lpat);
                    }
                } catch (...) {
                    pat_status_[k] = kBadInput;  // build_solvers() falls back to the full analysis
                }
            });
    }

    // ---- where every entry of A lands in L (C(lo,hi) -> L(hi,lo)); for duplicates the LAST one in the
    //      reference's traversal order wins (ereach assigns x[i] = a.x[p], lib.rs:2003) ---------------
    lap("pattern of L");
    std::vector<i64> Adst(annz_, -1);
    {
        std::atomic<bool> bad{false};
        parallel_ranges(n_, 1 << 16, [&](i64 h0, i64 h1) {
            std::vector<i64> last((size_t)n_, -1);  // (per thread: a column's entries are compared with each other only)
            for (i64 hi = h0; hi < h1; ++hi) {
                for (i64 q = C.p[hi]; q < C.p[hi + 1]; ++q) last[C.i[q]] = q;
                for (i64 q = C.p[hi]; q < C.p[hi + 1]; ++q) {
                    const i64 lo = C.i[q];
                    if (last[lo] != q) continue;  // overwritten by a later duplicate
                    i64 dst;
                    if (lo == hi) {
                        dst = Lp_[lo];
                    } else {
                        const i64 *b = Li_.data() + Lp_[lo] + 1, *e = Li_.data() + Lp_[lo + 1];
                        const i64 *f = std::lower_bound(b, e, hi);
                        if (f == e || *f != hi) {
                            bad = true;
                            return;
                        }
                        dst = f - Li_.data();
                    }
                    Adst[source[q]] = dst;
                }
            }
        });
        if (bad) return kBadInput;
    }

    lap("scatter map");
    // ---- panels: fundamental supernodes cut to kPanelWidth columns ------------------------------
    const std::vector<i64> super = fundamental_supernodes(S.parent, S.cp);
    nsuper_ = (i64)super.size() - 1;
    std::vector<int32_t> c0;  // panel starts
    for (i64 k = 0; k < nsuper_; ++k)
        for (i64 a = super[k]; a < super[k + 1]; a += kPanelWidth) c0.push_back((int32_t)a);
    c0.push_back((int32_t)n_);
    npanels_ = (i64)c0.size() - 1;
    std::vector<int32_t> panel_of(n_), prow(npanels_);
    for (i64 s = 0; s < npanels_; ++s) {
        for (i64 j = c0[s]; j < c0[s + 1]; ++j) panel_of[j] = (int32_t)s;
        prow[s] = (int32_t)(Lp_[c0[s + 1]] - Lp_[c0[s + 1] - 1] - 1);
    }
    auto R = [&](i64 s) { return Li_.data() + Lp_[c0[s + 1] - 1] + 1; };  // rows below panel s

    // ---- levels of the panel tree ---------------------------------------------------------------
    std::vector<int32_t> plev(npanels_, 0);
    for (i64 s = 0; s < npanels_; ++s)
        for (i64 j = c0[s]; j < c0[s + 1]; ++j) {
            const i64 p = S.parent[j];
            if (p >= 0 && panel_of[p] != s) plev[panel_of[p]] = std::max(plev[panel_of[p]], plev[s] + 1);
        }
    nlevels_ = 0;
    for (i64 s = 0; s < npanels_; ++s) nlevels_ = std::max<i64>(nlevels_, plev[s] + 1);
    std::vector<i64> lv_ptr(nlevels_ + 1, 0);
    for (i64 s = 0; s < npanels_; ++s) lv_ptr[plev[s] + 1]++;
    for (i64 l = 0; l < nlevels_; ++l) lv_ptr[l + 1] += lv_ptr[l];
    std::vector<int32_t> lv_panels(npanels_);
    {
        std::vector<i64> fill(lv_ptr.begin(), lv_ptr.end() - 1);
        for (i64 s = 0; s < npanels_; ++s) lv_panels[fill[plev[s]]++] = (int32_t)s;
    }

    // ---- update pairs (see chol_kernels.hip): inside a supernode panel by panel, between supernodes once per supernode
    struct Pair { int32_t d, s, q0; uint32_t mask; int32_t sup, t_begin, t_end, level, atomic; };
    std::vector<Pair> pairs;
    std::vector<int32_t> super_of_panel(npanels_), first_panel(nsuper_ + 1, 0);
    {
        i64 s = 0;
        for (i64 k = 0; k < nsuper_; ++k) {
            first_panel[k] = (int32_t)s;
            for (; s < npanels_ && c0[s] < super[k + 1]; ++s) super_of_panel[s] = (int32_t)k;
        }
        first_panel[nsuper_] = (int32_t)npanels_;
    }
    // Inside a supernode the panels are grouped in blocks of `bp`: a panel updates the later panels of ITS block at
    // once (32 columns at a time), the block as a whole updates the rest of the supernode when its last panel is done
    // (32 * bp columns accumulated in registers: fewer passes over the trailing part), the supernode as a whole updates
    // its ancestors when its last panel is done.
    const i64 bp = std::max<i64>(1, env_or("RSP_CHOL_BLOCK_PANELS", kCholBlockPanels));
    // tiny leaf supernodes (chol_tiny_panels): factor and update fused, four panels to a wavefront
    const bool use_tiny = !(getenv("RSP_CHOL_TINY") && getenv("RSP_CHOL_TINY")[0] == '0');
    // target-owned updates (chol_tile_update): no atomics, bit-reproducible factor
    grouped_ = env_or("RSP_CHOL_GROUPED", kCholGrouped) != 0;
    const bool grouped = grouped_;
    std::vector<char> tiny(npanels_, 0);
    if (use_tiny)
        for (i64 k = 0; k < nsuper_; ++k) {
            const i64 s = first_panel[k];
            if (first_panel[k + 1] - s == 1 && c0[s + 1] - c0[s] <= 4 && (c0[s + 1] - c0[s]) + prow[s] <= 16) tiny[s] = 1;
        }
    // (supernodes are independent here: contiguous ranges of them on the host threads, cut by the entries of L they hold, each
    // range into a list of its own; the lists are joined in order, so the pairs come out as one thread would produce them)
    const auto pairs_of = [&](i64 k0, i64 k1, std::vector<Pair> &out) {
        for (i64 k = k0; k < k1; ++k) {
            const i64 sc = super[k], se = super[k + 1];
            for (i64 d = first_panel[k]; d < first_panel[k + 1]; ++d) {
                if (tiny[d] && !grouped) continue;  // the tiny kernel applies its own updates, except in target-owned mode
                const i64 *Rd = R(d);
                const i64 rd = prow[d];
                const i64 blk0 = first_panel[k] + (d - first_panel[k]) / bp * bp;      // first panel of d's block
                const i64 blk1 = std::min<i64>(blk0 + bp, first_panel[k + 1]);          // one past its last panel
                const bool last_in_block = d + 1 == blk1, last = d + 1 == first_panel[k + 1];
                const i64 block_end_col = c0[blk1];  // == se for the last block
                for (i64 q = 0; q < rd;) {
                    const i64 row = Rd[q];
                    Pair p{(int32_t)d, panel_of[row], (int32_t)q, 0u, (int32_t)k, 0, 0, plev[d], 0};
                    if (row < block_end_col) {          // a later panel of the same block: this panel's columns
                        p.t_begin = (int32_t)(c0[d] - sc);
                        p.t_end = (int32_t)(c0[d + 1] - sc);
                    } else if (row < se) {              // the rest of the supernode: the whole block, once it is complete
                        if (!last_in_block) break;
                        p.t_begin = (int32_t)(c0[blk0] - sc);
                        p.t_end = (int32_t)(c0[blk1] - sc);
                    } else {                            // ancestors: the whole supernode, once it is complete
                        if (!last) break;
                        p.t_begin = 0;
                        p.t_end = (int32_t)(se - sc);
                        p.atomic = 1;
                    }
                    for (; q < rd && Rd[q] < c0[p.s + 1]; ++q) p.mask |= 1u << (Rd[q] - c0[p.s]);
                    out.push_back(p);
                }
            }
        }
    };
    {
        const i64 threads = std::max<i64>(1, std::min<i64>(host_threads(), nsuper_ >> 14));
        std::vector<i64> cut((size_t)threads + 1, nsuper_);
        cut[0] = 0;
        for (i64 t = 1; t < threads; ++t) {  // first supernode whose first entry of L lies at or beyond the t-th share
            const i64 want = lnz_ * t / threads;
            cut[(size_t)t] = std::lower_bound(super.begin(), super.end() - 1, want, [&](i64 col, i64 w) { return Lp_[col] < w; }) - super.begin();
        }
        std::vector<std::vector<Pair>> part((size_t)threads);
        std::vector<std::thread> pool;
        for (i64 t = 1; t < threads; ++t) pool.emplace_back(pairs_of, cut[(size_t)t], cut[(size_t)t + 1], std::ref(part[(size_t)t]));
        pairs_of(cut[0], cut[1], part[0]);
        for (std::thread &th : pool) th.join();
        size_t total = 0;
        for (const auto &v : part) total += v.size();
        pairs.reserve(total);
        for (const auto &v : part) pairs.insert(pairs.end(), v.begin(), v.end());
    }
    lap("panels, levels, pairs");
    std::vector<dev::CholTiny> tiny_items;
    std::vector<i64> tiny_tgt;
    std::vector<i64> tiny_lv_ptr(nlevels_ + 1, 0);
    std::vector<int32_t> tiny_incoming(nsuper_, -1);  // latest level at which a tiny panel updates supernode k
    {
        for (i64 s = 0; s < npanels_; ++s)
            if (tiny[s]) tiny_lv_ptr[plev[s] + 1]++;
        for (i64 l = 0; l < nlevels_; ++l) tiny_lv_ptr[l + 1] += tiny_lv_ptr[l];
        tiny_items.resize((size_t)tiny_lv_ptr[nlevels_]);
        std::vector<i64> fill(tiny_lv_ptr.begin(), tiny_lv_ptr.end() - 1);
        std::vector<i64> order((size_t)tiny_lv_ptr[nlevels_]);
        for (i64 s = 0; s < npanels_; ++s)
            if (tiny[s]) order[fill[plev[s]]++] = s;
        for (size_t k = 0; k < order.size(); ++k) {
            const i64 s = order[k];
            const i64 *Rs = R(s);
            const i64 r = prow[s];
            tiny_items[k] = dev::CholTiny{Lp_[c0[s]], (i64)tiny_tgt.size(), c0[s + 1] - c0[s], (int32_t)r, c0[s], 0};
            for (i64 q = 0; q < (grouped ? 0 : r); ++q)
                for (i64 qq = 0; qq <= q; ++qq) {
                    const i64 rowi = Rs[q], col = Rs[qq];
                    const i64 sp = panel_of[col], t = col - c0[sp], wsp = c0[sp + 1] - c0[sp];
                    i64 pos;
                    if (rowi < c0[sp + 1]) {
                        pos = rowi - c0[sp];
                    } else {
                        const i64 *Rb = R(sp), *Re = Rb + prow[sp];
                        const i64 *f = std::lower_bound(Rb, Re, rowi);
                        if (f == Re || *f != rowi) return kBadInput;  // parent / cp do not describe this pattern
                        pos = wsp + (f - Rb);
                    }
                    tiny_tgt.push_back(Lp_[col] + pos - t);
                    tiny_incoming[super_of_panel[sp]] = std::max(tiny_incoming[super_of_panel[sp]], plev[s]);
                }
        }
    }
    {   // plain stores inside a supernode are safe only if no other supernode still updates it while its panels are
        // being factored (true for fundamental supernodes: every other descendant hangs below the FIRST column)
        std::vector<int32_t> incoming(tiny_incoming);
        for (const Pair &p : pairs)
            if (p.atomic) incoming[super_of_panel[p.s]] = std::max(incoming[super_of_panel[p.s]], p.level);
        for (Pair &p : pairs)
            if (!p.atomic && incoming[p.sup] >= plev[first_panel[p.sup]]) p.atomic = 1;
    }
    const i64 npairs = (i64)pairs.size();
    if (npairs >= INT32_MAX) return kBadInput;
    // position in the target panel of every row a pair updates: rel[rel_off + (q - q0)], q = q0 .. |R_d| - 1
    std::vector<i64> rel_off(npairs + 1, 0);
    for (i64 k = 0; k < npairs; ++k) rel_off[k + 1] = rel_off[k] + (prow[pairs[k].d] - pairs[k].q0);
    std::vector<int32_t> rel((size_t)rel_off[npairs]);
    {
        std::vector<i64> tgt_ptr(npanels_ + 1, 0);
        for (const Pair &p : pairs) tgt_ptr[p.s + 1]++;
        for (i64 s = 0; s < npanels_; ++s) tgt_ptr[s + 1] += tgt_ptr[s];
        std::vector<int32_t> by_target(npairs);
        std::vector<i64> fill(tgt_ptr.begin(), tgt_ptr.end() - 1);
        for (i64 k = 0; k < npairs; ++k) by_target[fill[pairs[k].s]++] = (int32_t)k;
        // target panels are independent of each other: contiguous ranges of them on the host threads, cut by the rows they look up
        std::vector<i64> work((size_t)npanels_ + 1, 0);
        for (i64 s = 0; s < npanels_; ++s) {
            i64 rows = 0;
            for (i64 k = tgt_ptr[s]; k < tgt_ptr[s + 1]; ++k) rows += rel_off[by_target[k] + 1] - rel_off[by_target[k]];
            work[(size_t)s + 1] = work[(size_t)s] + rows + (rows ? (c0[s + 1] - c0[s]) + prow[s] : 0);
        }
        const i64 threads = std::max<i64>(1, std::min<i64>(host_threads(), work[(size_t)npanels_] >> 20));
        std::atomic<bool> bad{false};
        const auto positions = [&](i64 s0, i64 s1) {
            std::vector<int32_t> posmap((size_t)n_, -1);
            for (i64 s = s0; s < s1; ++s) {
                if (tgt_ptr[s] == tgt_ptr[s + 1]) continue;
                const i64 w = c0[s + 1] - c0[s];
                for (i64 t = 0; t < w; ++t) posmap[c0[s] + t] = (int32_t)t;
                const i64 *Rs = R(s);
                for (i64 k = 0; k < prow[s]; ++k) posmap[Rs[k]] = (int32_t)(w + k);
                for (i64 k = tgt_ptr[s]; k < tgt_ptr[s + 1]; ++k) {
                    const i64 pk = by_target[k];
                    const Pair &p = pairs[pk];
                    const i64 *Rd = R(p.d);
                    for (i64 q = p.q0; q < prow[p.d]; ++q) {
                        const int32_t pos = posmap[Rd[q]];
                        // every row below a column of L(:, d) that lies in panel s is a row of panel s (column structure
                        // of a Cholesky factor); anything else means parent/cp do not describe this pattern
                        if (pos < 0 || (Rd[q] < c0[s + 1] ? (Rd[q] < c0[s] || pos != Rd[q] - c0[s]) : (pos < w || Rs[pos - w] != Rd[q]))) {
                            bad = true;
                            return;
                        }
                        rel[(size_t)(rel_off[pk] + (q - p.q0))] = pos;
                    }
                }
            }
        };
        std::vector<std::thread> pool;
        std::vector<i64> cut((size_t)threads + 1, npanels_);
        cut[0] = 0;
        for (i64 t = 1; t < threads; ++t)
            cut[(size_t)t] = std::lower_bound(work.begin(), work.end() - 1, work[(size_t)npanels_] * t / threads) - work.begin();
        for (i64 t = 1; t < threads; ++t) pool.emplace_back(positions, cut[(size_t)t], cut[(size_t)t + 1]);
        positions(cut[0], cut[1]);
        for (std::thread &th : pool) th.join();
        if (bad) return kBadInput;
    }
    lap("tiny tables, row positions");
    // ---- target-owned mode: contributions regrouped by (level, target panel, 16-row tile) -----------------------------
    std::vector<dev::CholSub> subs;
    std::vector<dev::CholGroup> groups;
    std::vector<int32_t> group_level;
    std::vector<char> group_long;  // some source range of the group has more than 4 columns
    if (grouped) {
        struct Key { int32_t level, s, r16, pair, qa; };
        // in generation order: pairs ascending, rows ascending (counted, then written, both on the host threads)
        std::vector<i64> gen_off((size_t)npairs + 1, 0);
        parallel_ranges(npairs, 1 << 14, [&](i64 k0, i64 k1) {
            for (i64 k = k0; k < k1; ++k) {
                const int32_t *rp = rel.data() + rel_off[k];
                const i64 cnt = rel_off[k + 1] - rel_off[k];
                i64 tiles = cnt > 0;
                for (i64 e = 1; e < cnt; ++e) tiles += (rp[e] >> 4) != (rp[e - 1] >> 4);
                gen_off[(size_t)k + 1] = tiles;
            }
        });
        for (i64 k = 0; k < npairs; ++k) gen_off[(size_t)k + 1] += gen_off[(size_t)k];
        RawVector<Key> gen((size_t)gen_off[(size_t)npairs]);
        parallel_ranges(npairs, 1 << 14, [&](i64 k0, i64 k1) {
            for (i64 k = k0; k < k1; ++k) {
                const Pair &p = pairs[k];
                const int32_t *rp = rel.data() + rel_off[k];
                const i64 cnt = rel_off[k + 1] - rel_off[k];
                Key *out = gen.data() + gen_off[(size_t)k];
                for (i64 e = 0; e < cnt; ++e)
                    if (e == 0 || (rp[e] >> 4) != (rp[e - 1] >> 4)) *out++ = Key{p.level, p.s, rp[e] >> 4, (int32_t)k, (int32_t)(p.q0 + e)};
            }
        });
        lap("  tiles: keys");
        if (gen.size() >= (size_t)UINT32_MAX) return kBadInput;
        // order: level, target panel, tile, then generation order (which fixes the order of every sum).  Bucket by
        // level first, then sort (panel, tile, ordinal) triples packed in two words inside each bucket.
        std::vector<i64> lv_fill(nlevels_ + 1, 0);
        for (const Key &key : gen) lv_fill[key.level + 1]++;
        for (i64 l = 0; l < nlevels_; ++l) lv_fill[l + 1] += lv_fill[l];
        std::vector<std::pair<uint64_t, uint32_t>> order(gen.size());
        {
            std::vector<i64> at(lv_fill.begin(), lv_fill.end() - 1);
            for (size_t x = 0; x < gen.size(); ++x)
                order[at[gen[x].level]++] = {((uint64_t)(uint32_t)gen[x].s << 32) | (uint32_t)gen[x].r16, (uint32_t)x};
        }
        lap("  tiles: buckets");
        for (i64 l = 0; l < nlevels_; ++l) parallel_sort(order.begin() + lv_fill[l], order.begin() + lv_fill[l + 1]);
        lap("  tiles: sort");
        std::vector<Key> keys(gen.size());
        subs.resize(keys.size());
        // the records are independent of each other (host threads); the groups are formed in order afterwards
        parallel_ranges((i64)keys.size(), 1 << 16, [&](i64 xb, i64 xe) {
            for (i64 x = xb; x < xe; ++x) {
                keys[x] = gen[order[x].second];
                const Key &key = keys[x];
                const Pair &p = pairs[key.pair];
                const i64 sc = super[p.sup];
                dev::CholSub &u = subs[x];
                u.src_base = Lp_[sc];
                u.len_d = (int32_t)(Lp_[sc + 1] - Lp_[sc]);
                u.rs = prow[p.d];
                u.q0 = p.q0;
                u.mask = p.mask;
                u.t_begin = p.t_begin;
                u.t_end = p.t_end;
                u.qa = key.qa;
                u.pad = 0;
                u.pad2[0] = u.pad2[1] = 0;
                uint8_t map[16];
                std::fill(map, map + 16, (uint8_t)0xff);
                const int32_t *rp = rel.data() + rel_off[key.pair];
                for (i64 q = key.qa; q < prow[p.d] && (rp[q - p.q0] >> 4) == key.r16; ++q)
                    map[rp[q - p.q0] & 15] = (uint8_t)(q - key.qa);  // at most 16 rows share a tile
                std::memcpy(u.rowmap, map, 16);
            }
        });
        lap("  tiles: records");
        // groups: a new one wherever (level, panel, tile) changes -- marked and counted on the host threads, then filled
        const i64 nkeys = (i64)keys.size();
        const auto fresh_at = [&](i64 x) {
            return x == 0 || keys[x - 1].level != keys[x].level || keys[x - 1].s != keys[x].s || keys[x - 1].r16 != keys[x].r16;
        };
        const i64 gthreads = std::max<i64>(1, std::min<i64>(host_threads(), nkeys >> 16));
        std::vector<i64> gcut((size_t)gthreads + 1), gfirst((size_t)gthreads + 1, 0);
        for (i64 t = 0; t <= gthreads; ++t) gcut[(size_t)t] = nkeys * t / gthreads;
        {
            std::vector<std::thread> pool;
            const auto count = [&](i64 t) {
                i64 c = 0;
                for (i64 x = gcut[(size_t)t]; x < gcut[(size_t)t + 1]; ++x) c += fresh_at(x);
                gfirst[(size_t)t + 1] = c;
            };
            for (i64 t = 1; t < gthreads; ++t) pool.emplace_back(count, t);
            count(0);
            for (std::thread &th : pool) th.join();
        }
        for (i64 t = 0; t < gthreads; ++t) gfirst[(size_t)t + 1] += gfirst[(size_t)t];
        groups.resize((size_t)gfirst[(size_t)gthreads]);
        group_level.resize(groups.size());
        group_long.assign(groups.size(), 0);
        {
            std::vector<std::thread> pool;
            // a thread owns the groups that START in its range and follows the last of them into the next range
            const auto fill = [&](i64 t) {
                i64 g = gfirst[(size_t)t] - 1;
                i64 x = gcut[(size_t)t];
                while (x < gcut[(size_t)t + 1] && !fresh_at(x)) ++x;  // (the tail of a group that started earlier: its owner's)
                if (x >= gcut[(size_t)t + 1]) return;                 // no group starts in this range
                for (; x < nkeys; ++x) {
                    const bool fresh = fresh_at(x);
                    if (fresh && x >= gcut[(size_t)t + 1]) break;
                    const Key &key = keys[x];
                    const Pair &p = pairs[key.pair];
                    if (fresh) {
                        ++g;
                        groups[(size_t)g] = dev::CholGroup{Lp_[c0[p.s]], x, (c0[p.s + 1] - c0[p.s]) + prow[p.s], c0[p.s + 1] - c0[p.s], key.r16, 0};
                        group_level[(size_t)g] = key.level;
                    }
                    groups[(size_t)g].sub_count++;
                    if (p.t_end - p.t_begin > 4) group_long[(size_t)g] = 1;
                }
            };
            for (i64 t = 1; t < gthreads; ++t) pool.emplace_back(fill, t);
            fill(0);
            for (std::thread &th : pool) th.join();
        }
    }
    lap("target tiles");
    std::vector<dev::CholPair> pair_rec(grouped ? 0 : npairs);  // source-owned mode only
    for (i64 k = 0; k < (grouped ? 0 : npairs); ++k) {
        const Pair &p = pairs[k];
        dev::CholPair &r = pair_rec[k];
        const i64 sc = super[p.sup];
        r.src_base = Lp_[sc];
        r.tgt_base = Lp_[c0[p.s]];
        r.rel_off = rel_off[k];
        r.len_d = (int32_t)(Lp_[sc + 1] - Lp_[sc]);
        r.len_s = (c0[p.s + 1] - c0[p.s]) + prow[p.s];
        r.q0 = p.q0;
        r.mask = p.mask;
        r.rs = prow[p.d];
        r.t_begin = p.t_begin;
        r.t_end = p.t_end;
        r.atomic = p.atomic;
        r.pad[0] = r.pad[1] = 0;
    }
    // pairs by the level that issues them
    std::vector<i64> pair_lv_ptr(nlevels_ + 1, 0);
    for (const Pair &p : pairs) pair_lv_ptr[p.level + 1]++;
    for (i64 l = 0; l < nlevels_; ++l) pair_lv_ptr[l + 1] += pair_lv_ptr[l];
    std::vector<int32_t> pair_by_level(npairs);
    {
        std::vector<i64> fill(pair_lv_ptr.begin(), pair_lv_ptr.end() - 1);
        for (i64 k = 0; k < npairs; ++k) pair_by_level[fill[pairs[k].level]++] = (int32_t)k;
    }
    std::vector<dev::CholPanel> panel_rec(npanels_);
    ntall_ = 0;
    for (i64 s = 0; s < npanels_; ++s) {
        const int32_t dslot = prow[s] > 64 ? (int32_t)ntall_++ : -1;
        panel_rec[s] = dev::CholPanel{Lp_[c0[s]], c0[s + 1] - c0[s], prow[s], dslot, 0};
    }
    const i64 fuse_waves = env_or("RSP_CHOL_FUSE_WAVES", kCholFuseWaves);

    // ---- per level: factor items, row-solve items of the tall panels, update items of the level's panels ------
    levels_.assign(nlevels_, Level{0, 0, 0, 0, 0, 0, 0, 0, false});
    std::vector<int32_t> item_panel, item_block, multi, upd_pair, upd_q;
    std::vector<dev::CholGroup> final_groups;
    size_t next_group = 0;
    const bool short_groups = env_or("RSP_CHOL_SHORT_GROUPS", 1) != 0;
    for (i64 l = 0; l < nlevels_; ++l) {
        levels_[l].item_begin = (i64)item_panel.size();
        multi.clear();
        for (i64 k = lv_ptr[l]; k < lv_ptr[l + 1]; ++k) {
            const i64 s = lv_panels[k];
            if (tiny[s]) continue;
            // panels with <= 64 rows below the diagonal block are finished by one wavefront (block 0); larger ones
            // get a diagonal-only item (block -1) now and their row blocks in a second launch
            item_panel.push_back((int32_t)s);
            item_block.push_back(prow[s] <= 64 ? 0 : -1);
            if (prow[s] > 64) multi.push_back((int32_t)s);
        }
        levels_[l].item_count = (i64)item_panel.size() - levels_[l].item_begin;
        levels_[l].trsm_begin = (i64)item_panel.size();
        for (int32_t s : multi)
            for (i64 b = 0; b < (prow[s] + 63) / 64; ++b) {
                item_panel.push_back(s);
                item_block.push_back((int32_t)b);
            }
        levels_[l].trsm_count = (i64)item_panel.size() - levels_[l].trsm_begin;
        // few row blocks: one fused launch (every row-block wavefront factors the diagonal block itself) and the
        // parked diagonal blocks copied back by the update launch
        levels_[l].fused = levels_[l].trsm_count > 0 && levels_[l].trsm_count <= fuse_waves;
        if (levels_[l].fused) {
            const i64 first = levels_[l].item_begin, count = levels_[l].item_count;
            levels_[l].item_begin = (i64)item_panel.size();
            for (i64 k = first; k < first + count; ++k)
                if (item_block[k] == 0) {  // small panels as they were
                    item_panel.push_back(item_panel[k]);
                    item_block.push_back(0);
                }
            for (i64 k = levels_[l].trsm_begin; k < levels_[l].trsm_begin + levels_[l].trsm_count; ++k) {
                item_panel.push_back(item_panel[k]);
                item_block.push_back(item_block[k]);
            }
            levels_[l].item_count = (i64)item_panel.size() - levels_[l].item_begin;
            levels_[l].trsm_count = 0;
        }
        levels_[l].upd_begin = (i64)upd_pair.size();
        if (levels_[l].fused)
            for (int32_t s : multi) {
                upd_pair.push_back(~s);
                upd_q.push_back(0);
            }
        for (i64 k = pair_lv_ptr[l]; k < (grouped ? pair_lv_ptr[l] : pair_lv_ptr[l + 1]); ++k) {
            const i64 pk = pair_by_level[k];
            for (i64 q = pairs[pk].q0; q < prow[pairs[pk].d]; q += 16) {
                upd_pair.push_back((int32_t)pk);
                upd_q.push_back((int32_t)q);
            }
        }
        levels_[l].upd_count = (i64)upd_pair.size() - levels_[l].upd_begin;
        if (grouped) {  // the level's launch of chol_tile_update: parked diagonal blocks first, then its target tiles
            levels_[l].upd_begin = (i64)final_groups.size();
            if (levels_[l].fused)
                for (int32_t s : multi) final_groups.push_back(dev::CholGroup{0, 0, 0, 0, -1, s});
            for (; next_group < groups.size() && group_level[next_group] == l; ++next_group) {
                final_groups.push_back(groups[next_group]);
                if (!group_long[next_group] && short_groups) final_groups.back().w_s = -final_groups.back().w_s;  // fast path
            }
            levels_[l].upd_count = (i64)final_groups.size() - levels_[l].upd_begin;
        }
        levels_[l].tiny_begin = tiny_lv_ptr[l];
        levels_[l].tiny_count = tiny_lv_ptr[l + 1] - tiny_lv_ptr[l];
    }
    if (upd_pair.size() >= (size_t)INT32_MAX) return kBadInput;
    update_items_ = (i64)upd_pair.size();

    lap("records and work items");
    // ---- uploads ----------------------------------------------------------------------------------
    d_Lp.upload(Lp_);
    d_Lx.resize((size_t)lnz_);
    if (has_pinv_) d_pinv.upload(pinv_);
    d_Ax.resize((size_t)annz_);
    d_Adst.upload(Adst);
    d_panel_c0.upload(c0);
    d_panels.upload(panel_rec);
    d_pairs.upload(pair_rec);
    if (!grouped) d_rel.upload(rel);
    d_upd_pair.upload(upd_pair);
    d_upd_q.upload(upd_q);
    d_tiny.upload(tiny_items);
    d_subs.upload(subs);
    d_groups.upload(final_groups);
    d_tiny_tgt.upload(tiny_tgt);
    {
        std::vector<dev::CholFactorItem> items(item_panel.size());
        for (size_t k = 0; k < items.size(); ++k) items[k] = dev::CholFactorItem{panel_rec[item_panel[k]], item_block[k], c0[item_panel[k]]};
        d_items.upload(items);
    }
    d_status.resize(1);
    lap("uploads");
    return kOk;
}

int CholFactor::factorize(const CscView &A)
{
    if (A.n != n_ || A.nnz() != annz_) return kBadInput;
    int32_t bad = INT_MAX;
    factor_valid_ = false;
    replicas_.clear();  // (copies of the solve programmes on other devices: made again from the new values when next needed)
    factorize_values(A.x, 1, &bad);
    factor_valid_ = bad == INT_MAX;
    return factor_valid_ ? kOk : kNotPositiveDefinite;
}

// nsys systems with the analysed pattern and the values Ax_all[s * nnz(A) ...]; first_bad[s] = smallest column of
// system s with a pivot <= 0, or INT_MAX.  All systems go through the same launches (grid.y = system).
void CholFactor::factorize_values(const double *Ax_all, i64 nsys, int32_t *first_bad)
{
    if (nsys == 1 && lower_ && upper_) {
        solvers_stale_ = true;  // same pattern: the resident solve programmes only need the new values (build_solvers)
    } else {
        lower_.reset();
        upper_.reset();
    }
    batch_ = nsys;
    hipStream_t stream = nullptr;
    hipEvent_t ev0, ev1;
    RSP_HIP(hipEventCreate(&ev0));
    RSP_HIP(hipEventCreate(&ev1));
    d_Lx.resize((size_t)lnz_ * (size_t)nsys);
    d_Ax.resize((size_t)annz_ * (size_t)nsys);
    d_status.resize((size_t)nsys);
    if (annz_) RSP_HIP(hipMemcpyAsync(d_Ax.get(), Ax_all, (size_t)annz_ * (size_t)nsys * sizeof(double), hipMemcpyHostToDevice, stream));
    RSP_HIP(hipMemsetAsync(d_status.get(), 0x7f, (size_t)nsys * sizeof(int32_t), stream));  // 0x7f7f7f7f: "none"
    RSP_HIP(hipEventRecord(ev0, stream));
    RSP_HIP(hipMemsetAsync(d_Lx.get(), 0, (size_t)lnz_ * (size_t)nsys * sizeof(double), stream));
    const unsigned ny = (unsigned)nsys;
    if (annz_)
        hipLaunchKernelGGL(dev::chol_scatter_A, dim3((unsigned)((annz_ + 255) / 256), ny), dim3(256), 0, stream, d_Ax.get(),
                           d_Adst.get(), annz_, d_Lx.get(), annz_, lnz_);
    d_dbuf.resize((size_t)std::max<i64>(1, ntall_) * 1024 * (size_t)nsys);
    const i64 dstride = ntall_ * 1024;
    const dev::CholUpdateArgs U{d_upd_pair.get(), d_upd_q.get(), d_pairs.get(), d_rel.get(), d_Lx.get(), d_panels.get(), d_dbuf.get()};
    const dev::CholFactorArgs F{d_items.get(), d_Lx.get(), d_status.get(), d_dbuf.get()};
    const dev::CholTinyArgs T{d_tiny.get(), d_tiny_tgt.get(), d_Lx.get(), d_status.get()};
    const dev::CholTileArgs G{d_groups.get(), d_subs.get(), d_Lx.get(), d_panels.get(), d_dbuf.get()};
    for (const Level &lv : levels_) {
        if (lv.tiny_count)  // leaf supernodes of the level: factor + update in one kernel, 16 panels per workgroup
            hipLaunchKernelGGL(dev::chol_tiny_panels, dim3((unsigned)((lv.tiny_count + 15) / 16), ny), dim3(256), 0, stream, T,
                               lv.tiny_begin, lv.tiny_count, lnz_, grouped_ ? 0 : 1);
        if (lv.item_count)
            hipLaunchKernelGGL(dev::chol_panel_factor, dim3((unsigned)((lv.item_count + 3) / 4), ny), dim3(256), 0, stream, F,
                               lv.item_begin, lv.item_count, lv.fused ? 2 : 0, lnz_, dstride);
        if (lv.trsm_count)
            hipLaunchKernelGGL(dev::chol_panel_factor, dim3((unsigned)((lv.trsm_count + 3) / 4), ny), dim3(256), 0, stream, F,
                               lv.trsm_begin, lv.trsm_count, 1, lnz_, dstride);
        if (lv.upd_count && grouped_)  // target-owned tiles: plain stores, fixed order
            hipLaunchKernelGGL(dev::chol_tile_update, dim3((unsigned)((lv.upd_count + 3) / 4), ny), dim3(256), 0, stream, G,
                               lv.upd_begin, lv.upd_count, lnz_, dstride);
        else if (lv.upd_count)  // the level's panels are final: subtract their outer products from their ancestors
            hipLaunchKernelGGL(dev::chol_rank_update, dim3((unsigned)((lv.upd_count + 3) / 4), ny), dim3(256), 0, stream, U,
                               lv.upd_begin, lv.upd_count, lnz_, dstride);
    }
    RSP_HIP(hipEventRecord(ev1, stream));
    RSP_HIP(hipGetLastError());
#ifdef RSP_CHOL_CLOCKS
    {
        unsigned long long clk[16];
        RSP_HIP(hipMemcpyFromSymbol(clk, HIP_SYMBOL(dev::g_chol_clk), sizeof(clk)));
        fprintf(stderr, "chol clocks (10 ns ticks) of the last chain-level factor launch:");
        for (int i = 1; i < 7; ++i) fprintf(stderr, " %lld", (long long)(clk[i] - clk[0]));
        fprintf(stderr, "  | update: %lld %lld subs %lld\n", (long long)(clk[9] - clk[8]), (long long)(clk[10] - clk[8]), (long long)clk[12]);
    }
#endif
    std::vector<int32_t> status((size_t)nsys);
    RSP_HIP(hipMemcpy(status.data(), d_status.get(), (size_t)nsys * sizeof(int32_t), hipMemcpyDeviceToHost));
    RSP_HIP(hipEventSynchronize(ev1));
    float ms = 0.f;
    RSP_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    factor_ms_ = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    for (i64 s = 0; s < nsys; ++s) first_bad[s] = status[s] >= n_ ? INT_MAX : status[s];
}

// cholsol solve phase for every system of the last factorize_values(): B is nsys x n on the host, overwritten.
void CholFactor::solve_batch(double *B, i64 nsys)
{
    if (nsys <= 0 || n_ == 0) return;
    if (nsys != batch_) throw DeviceError("solve_batch: not the batch that was factored");
    if (!d_row_ptr.size()) {  // index arrays of the two sweeps, built once per pattern
        std::vector<int32_t> li((size_t)lnz_), row_ptr(n_ + 1, 0);
        for (i64 p = 0; p < lnz_; ++p) li[p] = (int32_t)Li_[p];
        for (i64 j = 0; j < n_; ++j)
            for (i64 p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) row_ptr[Li_[p] + 1]++;
        for (i64 r = 0; r < n_; ++r) row_ptr[r + 1] += row_ptr[r];
        std::vector<int32_t> row_col((size_t)row_ptr[n_]), row_pos((size_t)row_ptr[n_]), fill(row_ptr.begin(), row_ptr.end() - 1);
        for (i64 j = 0; j < n_; ++j)  // columns ascending: every row's entries end up in ascending column order
            for (i64 p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) {
                const int32_t at = fill[Li_[p]]++;
                row_col[at] = (int32_t)j;
                row_pos[at] = (int32_t)p;
            }
        auto by_level = [&](const std::vector<int32_t> &lev, std::vector<int32_t> &ptr, std::vector<int32_t> &items) {
            int32_t nl = 0;
            for (int32_t l : lev) nl = std::max(nl, l + 1);
            ptr.assign(nl + 1, 0);
            for (int32_t l : lev) ptr[l + 1]++;
            for (int32_t l = 0; l < nl; ++l) ptr[l + 1] += ptr[l];
            items.resize(lev.size());
            std::vector<int32_t> f(ptr.begin(), ptr.end() - 1);
            for (size_t k = 0; k < lev.size(); ++k) items[f[lev[k]]++] = (int32_t)k;
            return nl;
        };
        std::vector<int32_t> lev(n_, 0), ptr, items;
        for (i64 r = 0; r < n_; ++r)
            for (int32_t e = row_ptr[r]; e < row_ptr[r + 1]; ++e) lev[r] = std::max(lev[r], lev[row_col[e]] + 1);
        nlev_l_ = by_level(lev, ptr, items);
        d_lev_ptr_l.upload(ptr);
        d_lev_rows_l.upload(items);
        std::fill(lev.begin(), lev.end(), 0);
        for (i64 j = n_ - 1; j >= 0; --j)
            for (i64 p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) lev[j] = std::max(lev[j], lev[Li_[p]] + 1);
        nlev_t_ = by_level(lev, ptr, items);
        d_lev_ptr_t.upload(ptr);
        d_lev_cols_t.upload(items);
        d_Li32.upload(li);
        d_row_ptr.upload(row_ptr);
        d_row_col.upload(row_col);
        d_row_pos.upload(row_pos);
        RSP_HIP(hipFuncSetAttribute((const void *)dev::chol_batch_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kBatchMaxN * 8)));
    }
    DeviceBuffer<double> dB((size_t)nsys * (size_t)n_);
    RSP_HIP(hipMemcpy(dB.get(), B, (size_t)nsys * (size_t)n_ * sizeof(double), hipMemcpyHostToDevice));
    const dev::CholBatchSolveArgs S{d_Lp.get(),        d_Li32.get(),       d_row_ptr.get(),    d_row_col.get(),
                                    d_row_pos.get(),   d_lev_ptr_l.get(),  d_lev_rows_l.get(), d_lev_ptr_t.get(),
                                    d_lev_cols_t.get(), has_pinv_ ? d_pinv.get() : nullptr, d_Lx.get(), dB.get(),
                                    lnz_,              (int32_t)n_,        nlev_l_,            nlev_t_};
    hipLaunchKernelGGL(dev::chol_batch_solve, dim3((unsigned)nsys), dim3(256), (size_t)n_ * 8, nullptr, S);
    RSP_HIP(hipGetLastError());
    RSP_HIP(hipMemcpy(B, dB.get(), (size_t)nsys * (size_t)n_ * sizeof(double), hipMemcpyDeviceToHost));
}

void CholFactor::download(i64 *Lp, i64 *Li, double *Lx) const
{
    std::copy(Lp_.begin(), Lp_.end(), Lp);
    std::copy(Li_.begin(), Li_.end(), Li);
    d_Lx.download(Lx, (size_t)lnz_);
}

void CholFactor::build_solvers()
{
    if (lower_ && !solvers_stale_) return;
    join_pattern_jobs();
    const double t0 = now_s();
    if (lower_ && upper_ && solvers_stale_) {
        // refactorised: the resident programmes take the new values straight from d_Lx (kernels; nothing crosses PCIe).
        // The debugging switches that run the plain level programmes need the values on the host: host path below.
        const auto on = [](const char *name, char v) { return getenv(name) && getenv(name)[0] == v; };
        const bool device_refill = !on("RSP_REFILL_DEVICE", '0') && !on("RSP_TRI_SINGLE", '1') && !on("RSP_TRI_BLOCKED", '0');
        const CscView Lpat{n_, n_, Lp_.data(), Li_.data(), nullptr};
        RSP_HIP(hipDeviceSynchronize());  // nothing may still be reading the coefficients that are about to change
        if (device_refill && lower_->refresh_device(Lpat, d_Lx.get(), nullptr) && upper_->refresh_device(Lpat, d_Lx.get(), nullptr)) {
            RSP_HIP(hipStreamSynchronize(nullptr));  // the solve that follows may run on any stream
            solvers_stale_ = false;
            if (getenv("RSP_TIMING")) fprintf(stderr, "chol solvers: new values into the resident programmes on the device %.4fs\n", now_s() - t0);
            return;
        }
    }
    if (pat_ahead_ && pat_lower_ && pat_upper_ && pat_status_[0] == kOk && pat_status_[1] == kOk) {
        // the programmes were emitted beside the factorisation: upload them and pour the factor in, all on the device
        std::unique_ptr<TriSolver> lo = std::move(pat_lower_), up = std::move(pat_upper_);
        pat_ahead_ = false;
        const CscView Lpat{n_, n_, Lp_.data(), Li_.data(), nullptr};
        if (lo->adopt_staged() && up->adopt_staged() && lo->refresh_device(Lpat, d_Lx.get(), nullptr) &&
            up->refresh_device(Lpat, d_Lx.get(), nullptr)) {
            RSP_HIP(hipStreamSynchronize(nullptr));
            lower_ = std::move(lo);
            upper_ = std::move(up);
            solvers_stale_ = false;
            if (getenv("RSP_TIMING")) fprintf(stderr, "chol solvers: programmes emitted ahead, upload + values on the device %.3fs\n", now_s() - t0);
            return;
        }
        // (anything else -- another engine selected, no recipe -- takes the full analysis with the values on the host)
    }
    pat_ahead_ = false;
    if (single_use_ && !lower_ && pat_lower_ && pat_upper_ && pat_status_[0] == kOk && pat_status_[1] == kOk && pat_lower_->rows_prepared() &&
        pat_upper_->rows_prepared()) {
        // one vector, one use: the row kernel gathers the factor's values on the device; L never visits the host
        pat_lower_->rows_values_from_device(d_Lp.get(), d_Lx.get(), nullptr);
        pat_upper_->rows_values_from_device(d_Lp.get(), d_Lx.get(), nullptr);
        RSP_HIP(hipStreamSynchronize(nullptr));  // the solve that follows may run on any stream
        if (getenv("RSP_TIMING")) fprintf(stderr, "chol solvers: row kernel, values gathered on the device %.3fs\n", now_s() - t0);
        lower_ = std::move(pat_lower_);
        upper_ = std::move(pat_upper_);
        solvers_stale_ = false;
        return;
    }
    std::vector<double> Lx((size_t)lnz_);
    d_Lx.download(Lx.data(), Lx.size());
    const CscView L{n_, n_, Lp_.data(), Li_.data(), Lx.data()};
    if (lower_ && upper_ && solvers_stale_) {
        // refactorised: pour the new values into the resident programmes (both triangles side by side on the host), no
        // analysis and no emission; anything the recipe does not cover falls through to the full analysis below
        RSP_HIP(hipDeviceSynchronize());  // nothing may still be reading the coefficients that are about to change
        bool ok_lower = false, ok_upper = false;
        std::exception_ptr eb;
        std::thread other([&] {
            try {
                ok_upper = upper_->refresh_host(L);
            } catch (...) {
                eb = std::current_exception();
            }
        });
        try {
            ok_lower = lower_->refresh_host(L);
        } catch (...) {
            other.join();
            throw;
        }
        other.join();
        if (eb) std::rethrow_exception(eb);
        if (ok_lower && ok_upper) {
            const double t1 = now_s();
            lower_->refresh_upload();
            upper_->refresh_upload();
            solvers_stale_ = false;
            if (getenv("RSP_TIMING"))
                fprintf(stderr, "chol solvers: new values into the resident programmes %.3fs (download + host refill) + %.3fs (upload)\n", t1 - t0,
                        now_s() - t1);
            return;
        }
        lower_.reset();
        upper_.reset();
    }
    solvers_stale_ = false;
    const bool refill = !single_use_ && !(getenv("RSP_REFILL") && getenv("RSP_REFILL")[0] == '0');
    if (pat_lower_ && pat_upper_ && pat_status_[0] == kOk && pat_status_[1] == kOk) {
        // the patterns were analysed beside the factorisation: add the values, then stage what the solve will run
        std::unique_ptr<TriSolver> lo = std::move(pat_lower_), up = std::move(pat_upper_);
        const bool single = single_use_;
        auto finish = [&L, single](TriSolver &t, std::exception_ptr &e) {
            try {
                t.set_values(L);
                if (single) t.set_single_use();
                if (!t.single_use_levels()) t.stage_blocked();
            } catch (...) {
                e = std::current_exception();
            }
        };
        std::exception_ptr ea, eb;
        std::thread other(finish, std::ref(*up), std::ref(eb));
        finish(*lo, ea);
        other.join();
        if (ea) std::rethrow_exception(ea);
        if (eb) std::rethrow_exception(eb);
        lower_ = std::move(lo);
        upper_ = std::move(up);
        return;
    }
    pat_lower_.reset();
    pat_upper_.reset();
    std::unique_ptr<TriSolver> lo(new TriSolver), up(new TriSolver);
    if (refill) {  // a handle that may be refactorised: keep what a refill of the programmes needs
        lo->keep_for_refill();
        up->keep_for_refill();
    }
    if (analyze_pair(*lo, L, 0, *up, L, 1, single_use_) != kOk) throw DeviceError("factor is not a valid triangular matrix");
    lower_ = std::move(lo);
    upper_ = std::move(up);
}

// cholsol's solve phase on a device block: P (ipvec), lsolve, ltsolve, P' (pvec) -- lib.rs:383-386
static void chol_solve_phase(TriSolver &lower, TriSolver &upper, const int64_t *d_pinv, DeviceBuffer<double> &work, i64 n, double *dB, int layout,
                             i64 ld, i64 nrhs, hipStream_t stream)
{
    const int64_t *pinv = d_pinv;
    const bool perm = pinv != nullptr;
    const bool vector = (nrhs == 1 && (layout == 0 || ld == 1));
    if (layout == 1 && !perm && !vector) {  // native layout, nothing to move
        solve_pair_device(lower, upper, dB, ld, nrhs, stream);  // lsolve, ltsolve (lib.rs:384-385)
        return;
    }
    if (vector && !perm) {  // a vector is its own n x 1 row-major block
        solve_pair_device(lower, upper, dB, 1, 1, stream);
        return;
    }
    work.resize((size_t)n * (size_t)((layout == 1 && !vector) ? ld : nrhs));
    double *W = work.get();
    const dim3 tgrid((unsigned)((n + 31) / 32), (unsigned)((nrhs + 31) / 32)), tblock(256);
    const dim3 pgrid((unsigned)((n + 3) / 4)), pblock(256);
    // x = P b : x[pinv[k]] = b[k]  (ipvec, lib.rs:383)
    const dim3 vgrid((unsigned)((n + 255) / 256));
    if (vector)
        hipLaunchKernelGGL(dev::permute_vector, vgrid, dim3(256), 0, stream, dB, W, n, pinv, 1);
    else if (layout == 0)
        hipLaunchKernelGGL(dev::colmajor_to_rowmajor, tgrid, tblock, 0, stream, dB, ld, W, nrhs, n, (int)nrhs, pinv, 1);
    else
        hipLaunchKernelGGL(dev::permute_rows, pgrid, pblock, 0, stream, dB, W, ld, n, (int)nrhs, pinv, 1);
    const i64 ldw = (layout == 0 || vector) ? nrhs : ld;
    solve_pair_device(lower, upper, W, ldw, nrhs, stream);  // x = L \ x, x = L' \ x  (lib.rs:384-385)
    // b = P' x : b[k] = x[pinv[k]]  (pvec, lib.rs:386)
    if (vector)
        hipLaunchKernelGGL(dev::permute_vector, vgrid, dim3(256), 0, stream, W, dB, n, pinv, 0);
    else if (layout == 0)
        hipLaunchKernelGGL(dev::rowmajor_to_colmajor, tgrid, tblock, 0, stream, W, nrhs, dB, ld, n, (int)nrhs, pinv, 0);
    else
        hipLaunchKernelGGL(dev::permute_rows, pgrid, pblock, 0, stream, W, dB, ld, n, (int)nrhs, pinv, 0);
    RSP_HIP(hipGetLastError());
}

void CholFactor::solve_device(double *dB, int layout, i64 ld, i64 nrhs, bool permute, hipStream_t stream)
{
    if (nrhs <= 0 || n_ == 0) return;
    build_solvers();
    chol_solve_phase(*lower_, *upper_, (permute && has_pinv_) ? d_pinv.get() : nullptr, work_, n_, dB, layout, ld, nrhs, stream);
}

void CholFactor::solve_host_sharded(double *B, i64 ldb, i64 nrhs, const std::vector<int> &devices)
{
    const i64 G = (i64)devices.size();
    if (nrhs <= 0 || n_ == 0) return;
    if (G <= 1 || nrhs < 2 * G) {
        solve_host(B, ldb, nrhs);
        return;
    }
    // the programmes must be resident and idle on this handle's device before anything is copied from them
    build_solvers();
    const i64 widest = (nrhs + G - 1) / G;
    lower_->prepare_blocked(widest);
    upper_->prepare_blocked(widest);
    RSP_HIP(hipDeviceSynchronize());
    if ((i64)replicas_.size() < G) replicas_.resize((size_t)G);
    std::vector<std::string> errors((size_t)G);
    auto share = [&](i64 g) {
        try {
            const i64 lo = nrhs * g / G, hi = nrhs * (g + 1) / G, cols = hi - lo;
            if (cols == 0) return;
            RSP_HIP(hipSetDevice(devices[(size_t)g]));
            TriSolver *lo_s = lower_.get(), *up_s = upper_.get();
            const int64_t *pinv = has_pinv_ ? d_pinv.get() : nullptr;
            DeviceBuffer<double> *work = &work_;
            if (!(g == 0 && devices[0] == device_)) {  // everyone but the handle's own device (first in the list) works on a replica
                std::unique_ptr<Replica> &R = replicas_[(size_t)g];
                if (!R || R->device != devices[(size_t)g]) {
                    R.reset(new Replica);
                    R->device = devices[(size_t)g];
                    if (!lower_->clone_resident(R->lower, device_, R->device) || !upper_->clone_resident(R->upper, device_, R->device))
                        throw DeviceError("this handle's solve programmes cannot be replicated (not the blocked programme)");
                    if (has_pinv_) {
                        R->pinv.resize(d_pinv.size());
                        RSP_HIP(hipMemcpyPeer(R->pinv.get(), R->device, d_pinv.get(), device_, d_pinv.bytes()));
                    }
                    RSP_HIP(hipDeviceSynchronize());
                }
                lo_s = &R->lower;
                up_s = &R->upper;
                pinv = has_pinv_ ? R->pinv.get() : nullptr;
                work = &R->work;
            }
            DeviceBuffer<double> d((size_t)n_ * (size_t)cols);
            copy_block_to_device(d.get(), B + lo * ldb, n_, ldb, cols);
            hipStream_t stream = nullptr;
            RSP_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));  // shares of one device (a list that repeats it) run side by side
            try {
                chol_solve_phase(*lo_s, *up_s, pinv, *work, n_, d.get(), 0, n_, cols, stream);
                RSP_HIP(hipStreamSynchronize(stream));
            } catch (...) {
                (void)hipStreamDestroy(stream);
                throw;
            }
            (void)hipStreamDestroy(stream);
            copy_block_to_host(B + lo * ldb, d.get(), n_, ldb, cols);
        } catch (const std::exception &e) {
            errors[(size_t)g] = e.what()[0] ? e.what() : "error";
        }
    };
    int before = 0;
    RSP_HIP(hipGetDevice(&before));
    std::vector<std::thread> pool;
    for (i64 g = 1; g < G; ++g) pool.emplace_back(share, g);
    share(0);
    for (std::thread &t : pool) t.join();
    (void)hipSetDevice(before);
    for (const std::string &e : errors)
        if (!e.empty()) throw DeviceError("sharded solve: " + e);
}

void CholFactor::solve_host(double *B, i64 ldb, i64 nrhs)
{
    if (nrhs <= 0 || n_ == 0) return;
    DeviceBuffer<double> d((size_t)n_ * (size_t)nrhs);
    copy_block_to_device(d.get(), B, n_, ldb, nrhs);
    const bool timing = getenv("RSP_TIMING") != nullptr;
    const auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (timing) build_solvers();  // so that the figure below is the solve alone
    if (timing) RSP_HIP(hipDeviceSynchronize());
    const double t0 = clock();
    solve_device(d.get(), 0, n_, nrhs, true, nullptr);
    RSP_HIP(hipDeviceSynchronize());
    if (lower_) lower_->check_rows_run();  // (single-use handles: the flagged row kernel reports an abandoned wait here)
    if (upper_) upper_->check_rows_run();
    if (timing) fprintf(stderr, "chol solve: %lld right-hand side(s) on the device %.3fs\n", (long long)nrhs, clock() - t0);
    copy_block_to_host(B, d.get(), n_, ldb, nrhs);
}

void CholFactor::info(i64 *out) const
{
    out[0] = n_;
    out[1] = lnz_;
    out[2] = nsuper_;
    out[3] = nlevels_;
    out[4] = (i64)flops_;
    out[5] = annz_ * 12 + lnz_ * 20;  // bytes_chol (upper part of A counted as all given entries)
    out[6] = out[7] = out[8] = out[9] = 0;  // zeros until a solve has built the programmes
    if (lower_) lower_->summary(out[6], out[7]);
    if (upper_) upper_->summary(out[8], out[9]);
}

}  // namespace rsp
