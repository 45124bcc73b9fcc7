// CholFactor: host-side construction of the panel/update programme, GPU numeric factorisation, and the
// cholsol solve phase (P, lsolve, ltsolve, P') on the device.
#include <algorithm>
#include <climits>

#include "chol_engine.hpp"
#include "chol_kernels.hip"

namespace rsp {

int CholFactor::analyze(const CscView &A, const Symbolic &S)
{
    n_ = A.n;
    annz_ = A.nnz();
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

    // ---- where every entry of A lands in L (C(lo,hi) -> L(hi,lo)); for duplicates the LAST one in the
    //      reference's traversal order wins (ereach assigns x[i] = a.x[p], lib.rs:2003) ---------------
    std::vector<i64> Adst(annz_, -1);
    {
        std::vector<i64> last(n_, -1);
        for (i64 hi = 0; hi < n_; ++hi) {
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
                    if (f == e || *f != hi) return kBadInput;
                    dst = f - Li_.data();
                }
                Adst[source[q]] = dst;
            }
        }
    }

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

    // ---- pairs (source panel d -> target panel s), bucketed by target ---------------------------
    struct Pair { int32_t d, s, q0; uint32_t mask; };
    std::vector<Pair> pairs;
    for (i64 d = 0; d < npanels_; ++d) {
        const i64 *Rd = R(d);
        const i64 rd = prow[d];
        for (i64 q = 0; q < rd;) {
            const int32_t s = panel_of[Rd[q]];
            Pair p{(int32_t)d, s, (int32_t)q, 0u};
            for (; q < rd && Rd[q] < c0[s + 1]; ++q) p.mask |= 1u << (Rd[q] - c0[s]);
            pairs.push_back(p);
        }
    }
    const i64 npairs = (i64)pairs.size();
    if (npairs >= INT32_MAX) return kBadInput;
    std::vector<i64> tgt_ptr(npanels_ + 1, 0);
    for (const Pair &p : pairs) tgt_ptr[p.s + 1]++;
    for (i64 s = 0; s < npanels_; ++s) tgt_ptr[s + 1] += tgt_ptr[s];
    std::vector<int32_t> by_target(npairs);
    {
        std::vector<i64> fill(tgt_ptr.begin(), tgt_ptr.end() - 1);
        for (i64 k = 0; k < npairs; ++k) by_target[fill[pairs[k].s]++] = (int32_t)k;  // ascending d inside a target
    }

    // ---- slots (panel, position) and their tuple lists ------------------------------------------
    std::vector<i64> slot_base(npanels_ + 1, 0);
    for (i64 s = 0; s < npanels_; ++s) slot_base[s + 1] = slot_base[s] + (c0[s + 1] - c0[s]) + prow[s];
    const i64 nslots_all = slot_base[npanels_];
    std::vector<int32_t> tcount(nslots_all, 0), posmap(n_, -1);
    auto for_each_tuple = [&](auto &&fn) {
        for (i64 s = 0; s < npanels_; ++s) {
            if (tgt_ptr[s] == tgt_ptr[s + 1]) continue;
            const i64 w = c0[s + 1] - c0[s];
            for (i64 t = 0; t < w; ++t) posmap[c0[s] + t] = (int32_t)t;
            const i64 *Rs = R(s);
            for (i64 k = 0; k < prow[s]; ++k) posmap[Rs[k]] = (int32_t)(w + k);
            for (i64 k = tgt_ptr[s]; k < tgt_ptr[s + 1]; ++k) {
                const Pair &p = pairs[by_target[k]];
                const i64 *Rd = R(p.d);
                for (i64 q = p.q0; q < prow[p.d]; ++q) fn(s, slot_base[s] + posmap[Rd[q]], by_target[k], q);
            }
        }
    };
    i64 ntuples = 0;
    for_each_tuple([&](i64, i64 slot, int32_t, i64) { tcount[slot]++; ++ntuples; });
    // active slots in level order
    std::vector<int32_t> slot_panel, slot_pos, active(nslots_all, -1);
    std::vector<i64> slot_tptr(1, 0);
    levels_.assign(nlevels_, Level{0, 0, 0, 0, 0, 0});
    std::vector<int32_t> item_panel, item_block, multi;
    for (i64 l = 0; l < nlevels_; ++l) {
        levels_[l].slot_begin = (i64)slot_panel.size();
        levels_[l].item_begin = (i64)item_panel.size();
        multi.clear();
        for (i64 k = lv_ptr[l]; k < lv_ptr[l + 1]; ++k) {
            const i64 s = lv_panels[k];
            const i64 rows = slot_base[s + 1] - slot_base[s];
            for (i64 pos = 0; pos < rows; ++pos) {
                const i64 slot = slot_base[s] + pos;
                if (tcount[slot] == 0) continue;
                active[slot] = (int32_t)slot_panel.size();
                slot_panel.push_back((int32_t)s);
                slot_pos.push_back((int32_t)pos);
                slot_tptr.push_back(slot_tptr.back() + tcount[slot]);
            }
            // panels with <= 64 rows below the diagonal block are finished by one wavefront (block 0); larger ones
            // get a diagonal-only item (block -1) now and their row blocks in a second launch
            item_panel.push_back((int32_t)s);
            item_block.push_back(prow[s] <= 64 ? 0 : -1);
            if (prow[s] > 64) multi.push_back((int32_t)s);
        }
        levels_[l].slot_count = (i64)slot_panel.size() - levels_[l].slot_begin;
        levels_[l].item_count = (i64)item_panel.size() - levels_[l].item_begin;
        levels_[l].trsm_begin = (i64)item_panel.size();
        for (int32_t s : multi)
            for (i64 b = 0; b < (prow[s] + 63) / 64; ++b) {
                item_panel.push_back(s);
                item_block.push_back((int32_t)b);
            }
        levels_[l].trsm_count = (i64)item_panel.size() - levels_[l].trsm_begin;
    }
    std::vector<int32_t> tuple_pair(ntuples), tuple_q(ntuples);
    {
        std::vector<i64> fill(slot_tptr.begin(), slot_tptr.end() - 1);
        for_each_tuple([&](i64, i64 slot, int32_t pair, i64 q) {
            const i64 at = fill[active[slot]]++;
            tuple_pair[at] = pair;
            tuple_q[at] = (int32_t)q;
        });
    }
    std::vector<int32_t> pair_src(npairs), pair_q0(npairs);
    std::vector<uint32_t> pair_mask(npairs);
    for (i64 k = 0; k < npairs; ++k) {
        pair_src[k] = pairs[k].d;
        pair_q0[k] = pairs[k].q0;
        pair_mask[k] = pairs[k].mask;
    }

    // ---- uploads ----------------------------------------------------------------------------------
    d_Lp.upload(Lp_);
    d_Lx.resize((size_t)lnz_);
    if (has_pinv_) d_pinv.upload(pinv_);
    d_Ax.resize((size_t)annz_);
    d_Adst.upload(Adst);
    d_panel_c0.upload(c0);
    d_panel_rows.upload(prow);
    d_slot_panel.upload(slot_panel);
    d_slot_pos.upload(slot_pos);
    d_slot_tptr.upload(slot_tptr);
    d_tuple_pair.upload(tuple_pair);
    d_tuple_q.upload(tuple_q);
    d_pair_src.upload(pair_src);
    d_pair_q0.upload(pair_q0);
    d_pair_mask.upload(pair_mask);
    d_item_panel.upload(item_panel);
    d_item_block.upload(item_block);
    d_status.resize(1);
    return kOk;
}

int CholFactor::factorize(const CscView &A)
{
    if (A.n != n_ || A.nnz() != annz_) return kBadInput;
    lower_.reset();
    upper_.reset();
    hipStream_t stream = nullptr;
    hipEvent_t ev0, ev1;
    RSP_HIP(hipEventCreate(&ev0));
    RSP_HIP(hipEventCreate(&ev1));
    if (annz_) RSP_HIP(hipMemcpyAsync(d_Ax.get(), A.x, (size_t)annz_ * sizeof(double), hipMemcpyHostToDevice, stream));
    const int32_t none = INT_MAX;
    RSP_HIP(hipMemcpyAsync(d_status.get(), &none, sizeof(none), hipMemcpyHostToDevice, stream));
    RSP_HIP(hipEventRecord(ev0, stream));
    RSP_HIP(hipMemsetAsync(d_Lx.get(), 0, (size_t)lnz_ * sizeof(double), stream));
    if (annz_)
        hipLaunchKernelGGL(dev::chol_scatter_A, dim3((unsigned)((annz_ + 255) / 256)), dim3(256), 0, stream, d_Ax.get(),
                           d_Adst.get(), annz_, d_Lx.get());
    const dev::CholUpdateArgs U{d_slot_panel.get(), d_slot_pos.get(), d_slot_tptr.get(), d_tuple_pair.get(),
                                d_tuple_q.get(),    d_pair_src.get(), d_pair_q0.get(),   d_pair_mask.get(),
                                d_panel_c0.get(),   d_Lp.get(),       d_Lx.get()};
    const dev::CholFactorArgs F{d_item_panel.get(), d_item_block.get(), d_panel_c0.get(), d_panel_rows.get(),
                                d_Lp.get(),         d_Lx.get(),         d_status.get()};
    for (const Level &lv : levels_) {
        if (lv.slot_count)
            hipLaunchKernelGGL(dev::chol_panel_update, dim3((unsigned)((lv.slot_count + 3) / 4)), dim3(256), 0, stream, U,
                               lv.slot_begin, lv.slot_count);
        if (lv.item_count)
            hipLaunchKernelGGL(dev::chol_panel_factor, dim3((unsigned)((lv.item_count + 3) / 4)), dim3(256), 0, stream, F,
                               lv.item_begin, lv.item_count, 0);
        if (lv.trsm_count)
            hipLaunchKernelGGL(dev::chol_panel_factor, dim3((unsigned)((lv.trsm_count + 3) / 4)), dim3(256), 0, stream, F,
                               lv.trsm_begin, lv.trsm_count, 1);
    }
    RSP_HIP(hipEventRecord(ev1, stream));
    RSP_HIP(hipGetLastError());
    int32_t status = INT_MAX;
    RSP_HIP(hipMemcpy(&status, d_status.get(), sizeof(status), hipMemcpyDeviceToHost));
    RSP_HIP(hipEventSynchronize(ev1));
    float ms = 0.f;
    RSP_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    factor_ms_ = ms;
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    return status == INT_MAX ? kOk : kNotPositiveDefinite;
}

void CholFactor::download(i64 *Lp, i64 *Li, double *Lx) const
{
    std::copy(Lp_.begin(), Lp_.end(), Lp);
    std::copy(Li_.begin(), Li_.end(), Li);
    d_Lx.download(Lx, (size_t)lnz_);
}

void CholFactor::build_solvers()
{
    if (lower_) return;
    std::vector<double> Lx((size_t)lnz_);
    d_Lx.download(Lx.data(), Lx.size());
    const CscView L{n_, n_, Lp_.data(), Li_.data(), Lx.data()};
    std::unique_ptr<TriSolver> lo(new TriSolver), up(new TriSolver);
    if (lo->analyze(L, 0) != kOk || up->analyze(L, 1) != kOk) throw DeviceError("factor is not a valid triangular matrix");
    lower_ = std::move(lo);
    upper_ = std::move(up);
}

void CholFactor::solve_device(double *dB, int layout, i64 ld, i64 nrhs, bool permute, hipStream_t stream)
{
    if (nrhs <= 0 || n_ == 0) return;
    build_solvers();
    const bool perm = permute && has_pinv_;
    const int64_t *pinv = perm ? d_pinv.get() : nullptr;
    const bool vector = (nrhs == 1 && (layout == 0 || ld == 1));
    if (layout == 1 && !perm && !vector) {  // native layout, nothing to move
        solve_pair_device(*lower_, *upper_, dB, ld, nrhs, stream);  // lsolve, ltsolve (lib.rs:384-385)
        return;
    }
    work_.resize((size_t)n_ * (size_t)((layout == 1 && !vector) ? ld : nrhs));
    double *W = work_.get();
    const dim3 tgrid((unsigned)((n_ + 31) / 32), (unsigned)((nrhs + 31) / 32)), tblock(256);
    const dim3 pgrid((unsigned)((n_ + 3) / 4)), pblock(256);
    // x = P b : x[pinv[k]] = b[k]  (ipvec, lib.rs:383)
    if (layout == 0 || vector)
        hipLaunchKernelGGL(dev::colmajor_to_rowmajor, tgrid, tblock, 0, stream, dB, vector ? n_ : ld, W, nrhs, n_, (int)nrhs, pinv, 1);
    else
        hipLaunchKernelGGL(dev::permute_rows, pgrid, pblock, 0, stream, dB, W, ld, n_, (int)nrhs, pinv, 1);
    const i64 ldw = (layout == 0 || vector) ? nrhs : ld;
    solve_pair_device(*lower_, *upper_, W, ldw, nrhs, stream);  // x = L \ x, x = L' \ x  (lib.rs:384-385)
    // b = P' x : b[k] = x[pinv[k]]  (pvec, lib.rs:386)
    if (layout == 0 || vector)
        hipLaunchKernelGGL(dev::rowmajor_to_colmajor, tgrid, tblock, 0, stream, W, nrhs, dB, vector ? n_ : ld, n_, (int)nrhs, pinv, 0);
    else
        hipLaunchKernelGGL(dev::permute_rows, pgrid, pblock, 0, stream, W, dB, ld, n_, (int)nrhs, pinv, 0);
    RSP_HIP(hipGetLastError());
}

void CholFactor::solve_host(double *B, i64 ldb, i64 nrhs)
{
    if (nrhs <= 0 || n_ == 0) return;
    DeviceBuffer<double> d((size_t)n_ * (size_t)nrhs);
    RSP_HIP(hipMemcpy2D(d.get(), n_ * sizeof(double), B, ldb * sizeof(double), n_ * sizeof(double), nrhs, hipMemcpyHostToDevice));
    solve_device(d.get(), 0, n_, nrhs, true, nullptr);
    RSP_HIP(hipDeviceSynchronize());
    RSP_HIP(hipMemcpy2D(B, ldb * sizeof(double), d.get(), n_ * sizeof(double), n_ * sizeof(double), nrhs, hipMemcpyDeviceToHost));
}

void CholFactor::info(i64 *out) const
{
    out[0] = n_;
    out[1] = lnz_;
    out[2] = nsuper_;
    out[3] = nlevels_;
    out[4] = (i64)flops_;
    out[5] = annz_ * 12 + lnz_ * 20;  // bytes_chol (upper part of A counted as all given entries)
    out[6] = lower_ ? lower_->multi_program().nlevels : 0;
    out[7] = lower_ ? lower_->multi_program().ntasks : 0;
    out[8] = upper_ ? upper_->multi_program().nlevels : 0;
    out[9] = upper_ ? upper_->multi_program().ntasks : 0;
}

}  // namespace rsp
