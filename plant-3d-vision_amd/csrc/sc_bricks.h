// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- the dense stage on live bricks: brick_voxels, fills, unit verdicts, the confirm kernel, carve_brick_kernel, the light and the per-view kernels.

// Two views applied to the four voxels of a lane: both projections first, then the eight gathers of a
// lane in one flight (the kernels that call this wait on memory, not on arithmetic), then
// backprojection.c:79-83 for the first view and, for what it left alive, for the second.
// What a lane knows about its voxels is kept as WAVEFRONT masks in scalar registers (round 4, as in the survivor
// stages): alive[e] -- voxel e of the lane is not carved (:67) --, kept[e] -- some view so far has found it over
// foreground while it was alive (:81: a label 0 becomes 1; what the label was is looked at once, when the labels are
// written).  The comparisons of a projection leave lane masks anyway, so :79-83 is scalar and/or on them; the vector
// instructions of a (voxel, view) pair beyond its projection are the offset select, the gather and the bit test.
template <bool ALL_SAFE = false>
__device__ __forceinline__ void two_views(const ViewDesc &da, const ViewDesc &db, bool two, float x, float y,
                                          const float (&z)[4], unsigned long long (&alive)[4], unsigned long long (&kept)[4]) {
    const float aax = da.R[0] * x + da.R[1] * y, aay = da.R[3] * x + da.R[4] * y, aaz = da.R[6] * x + da.R[7] * y;
    const float bax = db.R[0] * x + db.R[1] * y, bay = db.R[3] * x + db.R[4] * y, baz = db.R[6] * x + db.R[7] * y;
    const uint32_t rowa = (uint32_t)da.strip, rowb = (uint32_t)db.strip;
    unsigned long long oka[4], okb[4];
    uint32_t wa[4], wb[4];
    int sha[4], shb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int v;
        // (dead lanes and lanes outside the picture gather word 0: cheaper than a branch around the load)
        bool ok = project<ALL_SAFE>(aax, aay, aaz, z[e], da, sha[e], v, oka[e]);
        wa[e] = load_mask_at(da.mask, ok ? mask_byte_offset(sha[e], v, rowa) : 0u);
        ok = project<ALL_SAFE>(bax, bay, baz, z[e], db, shb[e], v, okb[e]);
        wb[e] = load_mask_at(db.mask, ok ? mask_byte_offset(shb[e], v, rowb) : 0u);
    }
    const unsigned long long second = two ? ~0ull : 0ull;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned long long fg = __ballot(__builtin_amdgcn_ubfe(wa[e], (uint32_t)sha[e], 1u) != 0u);
        unsigned long long seen = oka[e] & alive[e];
        alive[e] &= ~(seen & ~fg);  // :79
        kept[e] |= seen & fg;       // :81
        fg = __ballot(__builtin_amdgcn_ubfe(wb[e], (uint32_t)shb[e], 1u) != 0u);
        seen = okb[e] & alive[e] & second;  // a voxel the first view carved is skipped (:67)
        alive[e] &= ~(seen & ~fg);
        kept[e] |= seen & fg;
    }
}

// The labels of a lane's four voxels behind two_views: carved -> -1; kept by some view and 0 before -> 1; else as before.
__device__ __forceinline__ void labels_behind(int32_t (&lab)[4], const unsigned long long (&alive)[4],
                                              const unsigned long long (&kept)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int32_t up = lab[e] == 0 ? 1 : lab[e];
        const int32_t live = __builtin_amdgcn_inverse_ballot_w64(kept[e]) ? up : lab[e];
        lab[e] = __builtin_amdgcn_inverse_ballot_w64(alive[e]) ? live : -1;
    }
}

template <bool FRESH, bool ALL_SAFE = false>  // ALL_SAFE: every view is certified (project())
__device__ __forceinline__ void brick_voxels(int32_t *__restrict__ labels, const GridDesc &g,
                                             const ViewDesc *__restrict__ views, int nviews,
                                             int32_t init, Append ap, uint32_t il, uint32_t j,
                                             uint32_t k0, uint32_t lb, uint32_t lane, uint32_t unit = 0,
                                             int nextra = 0) {  // views behind the `nviews` whose masks are packed too
    // bricks at the far y / z faces of the grid may stick out of it: lanes beyond ny or nz own
    // nothing (they still take part in the wave-wide ballots), a group at the end of a column
    // may be short (the pitch is a multiple of 64: groups are 16-byte aligned)
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    const uint64_t elem = ((uint64_t)il * g.ny + j) * g.nzp + k0;
    int32_t *p = labels + elem;
    int32_t lab[4], was[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) lab[e] = -1;  // what a lane does not own counts as carved
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) lab[e] = init;
    } else if (inside) {
        int4 q = *reinterpret_cast<const int4 *>(p);
        lab[0] = q.x; lab[1] = q.y; lab[2] = q.z; lab[3] = q.w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e >= nvalid) lab[e] = -1;  // row padding behind the last voxel
    }
    unsigned long long alive[4], kept[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        was[e] = lab[e];
        alive[e] = __ballot(lab[e] != -1);  // :67
        kept[e] = 0;
    }
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // :71, global plane index
    const float y = g.oy + (float)(int)j * g.vs;
    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(k0 + e) * g.vs;  // :73

    for (int vi = 0; vi < nviews; vi += 2) {
        if ((alive[0] | alive[1] | alive[2] | alive[3]) == 0) break;  // nothing left alive in this wavefront
        const bool two = vi + 1 < nviews;      // wave-uniform
        const ViewDesc da = views[vi];
        const ViewDesc db = views[two ? vi + 1 : vi];
        two_views<ALL_SAFE>(da, db, two, x, y, z, alive, kept);
    }
    // One more pair of views for a unit the dense views thinned out without emptying (32 .. 128 of its 256 voxels left):
    // masks that carve voxel by voxel (the noise scene: a quarter left after two views, a sixteenth after four), where a
    // survivor costs the lists more than a whole unit costs here.  A plant's units hold fewer, a solid's more.  The
    // survivor stage behind applies those two views again -- nothing changes the second time (a carve is final, a kept
    // 0 is already 1).
    if (nextra >= 2 && ap.list != nullptr) {  // wave-uniform
        const uint32_t left = (uint32_t)(__popcll(alive[0]) + __popcll(alive[1]) + __popcll(alive[2]) + __popcll(alive[3]));
        if (left >= 32u && left <= 128u) {
            const ViewDesc da = views[nviews];
            const ViewDesc db = views[nviews + 1];
            two_views<ALL_SAFE>(da, db, true, x, y, z, alive, kept);
        }
    }
    labels_behind(lab, alive, kept);

    const bool changed = FRESH || lab[0] != was[0] || lab[1] != was[1] || lab[2] != was[2] || lab[3] != was[3];
    if (inside && changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);

    if (ap.list != nullptr) {
        ap.sub = (lb * 0x9E3779B1u) >> 24;
        uint32_t total = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) total += (uint32_t)__popcll(alive[e]);
        bool bulked = false;
        if (ap.bulk != nullptr && total >= ap.bulk_min) {  // wave-uniform
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&ap.ctl->count[3][ap.sub].n, 1u);
            pos = __shfl(pos, 0);
            bulked = pos < ap.bulkcap;  // (a full sub-list: the voxels take the ordinary lists)
            if (bulked && lane == 0)
                ap.bulk[(size_t)ap.sub * ap.bulkcap + pos] = lb * 4u + unit;
        }
        if (total != 0 && !bulked) {  // wave-uniform
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ap.ctl->count[0][ap.sub].n, total);
            base = __shfl(base, 0);
            if (base + total > ap.subcap) {
                if (lane == 0) ap.ctl->overflow = 1u;
            } else {
                // In ADDRESS order (round 4): the survivors of lower lanes first, then the lane's own in z order -- lanes
                // 4 c .. 4 c + 3 hold 16 consecutive voxels of column c.  A chunk of the survivor stage behind is then
                // a run of neighbouring voxels: its lanes gather from the same mask words, and where most of them are
                // carved (the noise scene: 33 M) their -1 stores fill whole lines (element-major order -- voxel e of
                // every lane, then e + 1 -- made every such store a 4-byte piece of a 16-byte stride: 16 bytes of write
                // traffic per label).
                uint32_t *dst = ap.list + (size_t)ap.sub * ap.subcap + base;
                uint32_t rank = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) rank += lanes_below(alive[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (__builtin_amdgcn_inverse_ballot_w64(alive[e])) {
                        dst[rank] = (uint32_t)(elem + e) | (lab[e] == 0 ? 0x80000000u : 0u);
                        ++rank;
                    }
                }
            }
        }
    }
}

// The bricks of a strip the flags kernel has settled.  EMPTY (flag 1): live voxels become -1, dead
// ones are -1 already -- one 16-byte store per lane and brick, nothing else.  FULL (flag 2): every
// view keeps every voxel, so a label 0 becomes 1 and any other label stays (backprojection.c:81):
// `kept` is that value for a volume known to hold `init` everywhere (fresh), else the labels are
// read, patched and written back.
struct Fill {
    int32_t kept;   // label of a FULL brick's voxels when the volume is fresh: init == 0 ? 1 : init
    int32_t fresh;  // the volume holds `init` everywhere (nothing applied since clear)
    int32_t init;   // ... and this is what an UNTOUCHED brick (flag 6) of a fresh volume gets
};

// (f: the lane's flag of the strip -- lane b holds brick b's, see strip_flag: a persistent store block fetches the flags
// of several strips before it stores any of them, because the wait for a load is also a wait for every store issued
// before it: vector loads and stores retire through one in-order counter)
__device__ __forceinline__ uint32_t strip_flag(const uint8_t *__restrict__ flags, uint32_t strip, uint32_t bricks_z) {
    const uint32_t lane = threadIdx.x & 63;
    return (lane < bricks_z) ? flags[strip * bricks_z + lane] : 0u;  // (bricks_z <= 64: the brick form is for nz <= 4096)
}
__device__ __forceinline__ void store_culled_bricks(int32_t *__restrict__ labels, const GridDesc &g, uint32_t f, uint32_t strip,
                                                    uint32_t bricks_y, uint32_t bricks_z, Fill fill, bool prefilled = false) {  // prefilled: its labels are -1 already (SpecFill)
    typedef int v4i __attribute__((ext_vector_type(4)));
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t il = strip / bricks_y, by = strip - il * bricks_y;
    const uint32_t j0 = by * kBrickY + wave * 4, j = j0 + (lane >> 4);
    const unsigned long long culled = prefilled ? 0ull : __ballot(f == 1u), full = __ballot(f == 2u);
    // UNTOUCHED bricks (6) keep their labels: only a fresh volume, whose labels exist as `init` in name
    // only, has something to write there
    const unsigned long long untouched = fill.fresh ? __ballot(f == 6u) : 0ull;
    if (j >= g.ny) return;  // a strip at the far y face may stick out of the grid
    const uint32_t l16 = lane & 15u;
    const bool tail_ok = (bricks_z - 1u) * kBrickZ + l16 * 4u < g.nz;  // the last brick of a column may stick out of it
    const unsigned long long last = 1ull << (bricks_z - 1u);
    // EMPTY bricks, the bulk of the fill: a scalar walk over the set bits, and per brick ONE vector instruction -- the
    // store, its address the wavefront's (scalar) plus the lane's 32-bit offset within the wavefront's four columns.
    // (Round 4: the store blocks share their SIMDs with the survivor stages' wavefronts, whose vector instructions bound
    // those stages; the compiler's form of this loop -- 64-bit shifts of the masks per lane, a 64-bit address per lane
    // and turn -- was 5 M of the final stage's 34 M vector instructions.)
    {
        const char *wbase = reinterpret_cast<const char *>(labels) + ((uint64_t)il * g.ny + j0) * g.nzp * 4u;
        const uint32_t voff = (lane >> 4) * g.nzp * 4u + l16 * 16u;
        v4i minus = {-1, -1, -1, -1};
        asm volatile("" : "+v"(minus));  // four registers for the life of the loop
        unsigned long long body = culled & ~last;
        while (body != 0) {  // wave-uniform
            const uint32_t bz = (uint32_t)__builtin_ctzll(body);
            body &= body - 1;
            // streaming store: the fill is written once and not read again by this batch; kept
            // out of the caches it does not evict the masks the next batch packs
            asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(voff), "v"(minus), "s"(wbase + bz * 256u) : "memory");
        }
        if ((culled & last) != 0 && tail_ok)
            asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(voff), "v"(minus), "s"(wbase + (bricks_z - 1u) * 256u) : "memory");
        // a store of more than 8 bytes reads its data registers a little after it issues, and the compiler, which cannot
        // see that the statements above are stores, would be free to reuse them at once: they stay live over two more
        // wait states
        asm volatile("s_nop 1" ::"v"(minus));
    }
    unsigned long long rest = full | untouched;
    if (rest == 0) return;
    int32_t *col = labels + ((uint64_t)il * g.ny + j) * g.nzp;
    while (rest != 0) {  // rare: FULL bricks, and the UNTOUCHED ones of a fresh volume
        const uint32_t bz = (uint32_t)__builtin_ctzll(rest);
        rest &= rest - 1;
        const bool isfull = (full >> bz) & 1ull;
        const uint32_t k0 = bz * kBrickZ + l16 * 4;
        if (k0 >= g.nz) continue;
        if (!isfull || fill.fresh) {
            const int32_t val = isfull ? fill.kept : fill.init;
            const v4i vv = {val, val, val, val};
            __builtin_nontemporal_store(vv, reinterpret_cast<v4i *>(col + k0));  // (rows are whole 64-voxel bricks: nzp % 64 == 0)
        } else {  // FULL brick of a stored volume: 0 -> 1, the rest as it is
            const uint32_t n = min(4u, g.nz - k0);
            for (uint32_t e = 0; e < n; ++e)
                if (col[k0 + e] == 0) col[k0 + e] = 1;
        }
    }
}

#ifdef SC_TRACE_DENSE  // diagnostic builds only (tools/probes/dense_trace.py): what every walker wavefront of the dense
__device__ uint32_t g_dense_trace[8192 * 8];  // stage did (rows 0..4095)
#endif
// FULL candidates (flag 3 / 7: every view the flags kernel could see keeps the brick whole, but the masks
// of views [v0, v1) were packed only afterwards, beside the dense stage) put the question to those
// views.  The flags kernel's organisation for its own FULL rounds -- a brick per lane, a view per wavefront and
// round, verdicts joined in LDS -- over the CANDIDATE LIST that kernel leaves (ListCtl::ncand: eight sub-lists, one behind the other here), 64 entries per block:
// every lane has a question.  (Until round 4 a block took 64 bricks where they lie, candidates or not: five
// candidates cost the 7 rounds of 8 views that 64 cost, and the kernel is bound by what it issues, not by the chain
// of its rounds -- 65 us on a bulky object with 8 or with 16 wavefronts per block: DESIGN_APPENDIX 12.)
// Kept by all: flag 2 (or 6, unseen), filled like any FULL brick.
// Otherwise flag 5 and a place on the LATE list: the special kernel (carve_special_kernel) carves such a brick
// over all the views of the batch, unit by unit.
constexpr int kConfirmWaves = 8;  // (16: a solid object 0.219 -> 0.253 ms -- a round asks all its views about every brick still
                                  // a candidate, so wider rounds ask more views about bricks an earlier one would have dropped)
// Round 5 -- the road of a candidate that fails.  Until then it went on the late list and the special kernel took it
// through EVERY view of the batch, unit by unit, one wavefront per unit: a verdict round and a chain of two-view turns
// each, 1 374 bricks = 5 496 such chains on the reference's own configuration and the longest part of that kernel.
// Now (UnitRoad::on, when the batch has a bulk list and the dense stage's lists did not overflow) the brick's labels are
// written HERE -- every view the flags kernel could see keeps the brick whole, so they are what the dense stage would
// have left: `init`, or 1 over 0 where some view saw it -- and its four units join the bulk units: the special kernel
// asks the remaining views about each as a whole (unit_verdicts) and only the undecided ones project its voxels, as
// work items of the final stage, which has the wavefronts to hide their latency.  The other road stays for batches
// without a bulk list (SC_OPT_BULK_MIN 0, more than 128 views, masks without cell maps) and for a dense stage that
// overflowed.
struct UnitRoad {
    int32_t *labels;   // null: every failed candidate takes the late list
    int32_t init, fresh;
    uint32_t nbricks;  // the late list's room: unit-road bricks are entered from its far end
};

__global__ __launch_bounds__(64 * kConfirmWaves) void brick_confirm_kernel(
    GridDesc g, const ViewDesc *__restrict__ views, int v0, int v1, uint32_t bricks_y, uint32_t bricks_z,
    uint8_t *__restrict__ flags, const uint32_t *__restrict__ cands, uint32_t cand_per, uint32_t *__restrict__ late,
    ListCtl *ctl, uint32_t parity, UnitRoad ur) {
    if (v0 >= v1) return;  // no view was packed late
    // the sub-lists' groups of 64, one behind the other (a batch without open candidates: every block leaves here)
    uint32_t ncand[kCandSub], gfirst[kCandSub + 1];
    gfirst[0] = 0;
#pragma unroll
    for (int s = 0; s < kCandSub; ++s) {
        ncand[s] = ctl->ncand[parity][s].n;
        gfirst[s + 1] = gfirst[s] + ((ncand[s] + 63u) >> 6);
    }
    __shared__ unsigned long long s_full[kConfirmWaves], s_seen[kConfirmWaves];
    __shared__ unsigned long long s_road, s_rseen;  // the group's failed candidates that take the units' road, and who saw them
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t per_plane = bricks_y * bricks_z;
    // a persistent grid over the groups of 64 candidates
    for (uint32_t grp = blockIdx.x; grp < gfirst[kCandSub]; grp += gridDim.x) {
        uint32_t sub = 0, first = 0, n = ncand[0];
#pragma unroll
        for (int s = 1; s < kCandSub; ++s)
            if (grp >= gfirst[s]) { sub = (uint32_t)s; first = gfirst[s]; n = ncand[s]; }
        const uint32_t idx = (grp - first) * 64u + lane;
        const bool isc = idx < n;
        const uint32_t lb = isc ? cands[(size_t)sub * cand_per * 64u + idx] : 0u;
        const uint32_t fl = isc ? flags[lb] : 0u;
        unsigned long long any_seen = __ballot(fl == 3u);  // some view so far saw the brick whole (7: none sees it)
        unsigned long long cand = __ballot(isc);
        const uint32_t il = lb / per_plane;
        const uint32_t rem = lb - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
        for (int base = v0; base < v1 && cand != 0; base += kConfirmWaves) {  // block-uniform
            const int vi = base + (int)wave;
            bool keeps = true, sees = false;
            if (vi < v1 && ((cand >> lane) & 1ull)) {
                const ViewDesc d = views[vi];
                const uint32_t v = brick_verdict(d, g, x, (int)(by * kBrickY), (int)(bz * kBrickZ), d.tiles_x);
                keeps = v == 2u || v == 4u;
                sees = v == 2u;
            }
            const unsigned long long mf = __ballot(keeps), ms = __ballot(sees);
            __syncthreads();  // the previous round's masks have been read by everybody
            if (lane == 0) { s_full[wave] = mf; s_seen[wave] = ms; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kConfirmWaves; ++w) { cand &= s_full[w]; any_seen |= s_seen[w]; }
        }
        const bool road = ur.labels != nullptr && ctl->overflow == 0u;  // block-uniform (the dense stage set the flag)
        if (wave == 0) {
            if (isc) flags[lb] = ((cand >> lane) & 1ull) ? (((any_seen >> lane) & 1ull) ? 2 : 6) : 5;
            const bool failed = isc && !((cand >> lane) & 1ull);
            const unsigned long long m = __ballot(failed);
            if (m != 0) {
                uint32_t pos = 0;
                if (lane == 0) pos = atomicAdd(road ? &ctl->nlate_units : &ctl->nlate, (uint32_t)__popcll(m));
                pos = __shfl(pos, 0);
                const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                const uint32_t at = pos + (uint32_t)__popcll(m & below);
                if (failed) late[road ? ur.nbricks - 1u - at : at] = lb;  // (failed candidates are bricks: both ends fit)
            }
            if (lane == 0) { s_road = road ? m : 0ull; s_rseen = any_seen; }  // (words of their own: a wavefront may still be reading the last round's)
        }
        __syncthreads();
        if (road) {
            // the labels of the bricks that take the units' road, a brick per wavefront and turn: 16 columns x 64 voxels =
            // four 16-byte stores per lane (lane = (column & 3) * 16 + group of 4 voxels)
            const unsigned long long mr = s_road, ms = s_rseen;
            uint32_t nth = 0;
            for (unsigned long long mm = mr; mm != 0; mm &= mm - 1, ++nth) {
                if ((nth & (uint32_t)(kConfirmWaves - 1)) != wave) continue;
                const uint32_t src = (uint32_t)__builtin_ctzll(mm);
                const uint32_t b = __builtin_amdgcn_readlane(lb, src);
                const bool seen = (ms >> src) & 1ull;
                const uint32_t bil = b / per_plane, brem = b - bil * per_plane;
                const uint32_t bby = brem / bricks_z, bbz = brem - bby * bricks_z;
                const uint32_t k0 = bbz * kBrickZ + (lane & 15u) * 4u;
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) {
                    const uint32_t j = bby * kBrickY + i * 4u + (lane >> 4);
                    if (j >= g.ny || k0 >= g.nz) continue;
                    int32_t *p = ur.labels + ((uint64_t)bil * g.ny + j) * g.nzp + k0;  // the pitch is a multiple of 64: 16-byte groups
                    if (ur.fresh) {
                        const int32_t val = (seen && ur.init == 0) ? 1 : ur.init;  // backprojection.c:81 by a view that keeps the brick whole
                        *reinterpret_cast<int4 *>(p) = make_int4(val, val, val, val);
                    } else if (seen) {
                        int4 q = *reinterpret_cast<const int4 *>(p);
                        const bool changed = q.x == 0 || q.y == 0 || q.z == 0 || q.w == 0;
                        q.x = q.x == 0 ? 1 : q.x; q.y = q.y == 0 ? 1 : q.y; q.z = q.z == 0 ? 1 : q.z; q.w = q.w == 0 ? 1 : q.w;
                        if (changed) *reinterpret_cast<int4 *>(p) = q;
                    }
                }
            }
        }
        __syncthreads();  // everybody has read s_road / s_rseen before the next group's wave 0 writes them
    }
}

// The dense kernel proper: a persistent grid walks the live list, one brick per block and turn
// (wavefront w owns columns 4w..4w+3 of the brick); runs of kXcdRun consecutive entries
// (neighbouring bricks, which project onto the same mask lines) stay on one XCD.  Blocks behind
// the walkers, one per strip, fill the bricks found empty of strips [0, nstore) (the final list
// stage fills the others, see carve_list_kernel).
template <bool FRESH, bool ALL_SAFE = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(102))) void carve_brick_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                             const ViewDesc *__restrict__ views,
                                                             int nviews, int32_t init, Append ap,
                                                             uint32_t bricks_y, uint32_t bricks_z,
                                                             const uint8_t *__restrict__ flags,
                                                             const uint32_t *__restrict__ live,
                                                             ListCtl *ctl, uint32_t nwalkers,
                                                             uint32_t nstore, PackJob ride, int pack_rows,
                                                             uint32_t parity, int nverd_arg, uint32_t verd_max_live,
                                                             uint32_t bulk_min_live, int nextra) {
    // Behind the walkers come the riders, and the store blocks LAST: blocks start in the order of their numbers, and what
    // the riders pack is waited for by the next kernel, while a store only has to be done by the end of this one
    // (with the store blocks in front the riders started when the stores were through, and a sixteenth of the fill
    // cost this kernel the 5.4 us it takes on its own).
    const uint32_t nride = gridDim.x - nwalkers - nstore;
    if (blockIdx.x >= nwalkers && blockIdx.x < nwalkers + nride) {
        // riders: the masks of the views the later stages apply are packed here, beside the walkers
        // (this stage waits on gathers and arithmetic, the packing on HBM reads).  One short block per
        // panel: persistent riders measured the same or slower.
        const uint32_t b = blockIdx.x - nwalkers;
        if (pack_rows == 0) pack_band_block(ride, b);
        else if (pack_rows == 1) pack16_block<1>(ride, b);
        else if (pack_rows == 2) pack16_block<2>(ride, b);
        else if (pack_rows == 8) pack16_block<8>(ride, b);
        else pack16_block<4>(ride, b);
        return;
    }
    if (blockIdx.x >= nwalkers) {
        store_culled_bricks(labels, g, strip_flag(flags, blockIdx.x - nwalkers - nride, bricks_z), blockIdx.x - nwalkers - nride, bricks_y, bricks_z,
                            Fill{init == 0 ? 1 : init, FRESH ? 1 : 0, init});
        return;
    }
    // Walkers are WAVEFRONTS: each takes the next live brick of its XCD's runs (runs of kXcdRun consecutive entries --
    // neighbouring bricks, which project onto the same mask lines -- stay on one XCD; a ticket counter per XCD), asks
    // the views packed ahead about the brick's four UNITS (16 columns x 16 voxels) at the cell level, one (unit, view)
    // pair per lane, carves the units some view finds empty without projecting a voxel -- two thirds of a plant's:
    // the brick is live because a 32x32 tile under it touches the plant, the unit lies beside it -- and projects the
    // others.  (A block of four wavefronts per brick, one unit each, left three in four idle once units are culled;
    // tickets keep every wavefront busy whatever the bricks hold.)
    const uint32_t nlive = ctl->nlive[parity];
    // masks whose tiles settled less than half of the bricks (noise: none) have no structure for the cells to find
    const int nverd = nlive <= verd_max_live ? nverd_arg : 0;
    // ... and a batch whose tiles left only a sliver of the bricks live is a thin object: its few bulky units (a plant's
    // 3 268 of 30 308) are not worth a list of their own -- their voxels go with the others (decided here, on the
    // device, from this batch's own live count)
    if (nlive < bulk_min_live) ap.bulk = nullptr;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t xcd = blockIdx.x & 7u;
    // (the first ticket of a wavefront is its own number among the XCD's: a thousand atomics on one address at the
    // kernel's start would take longer than the first bricks)
    const uint32_t per_xcd = (nwalkers >> 3) * (kBlock / 64);
    bool first = true;
    uint32_t misses = 0, turn = 0;
#ifdef SC_TRACE_DENSE
    const uint64_t tr0 = wall_clock64();
    uint32_t tr_bricks = 0, tr_units = 0, tr_verd = 0, tr_unit = 0, tr_tick = 0;
#endif
    for (;;) {
#ifdef SC_TRACE_DENSE
        const uint64_t tra = wall_clock64();
#endif
        uint32_t t = (blockIdx.x >> 3) * (kBlock / 64) + (threadIdx.x >> 6);
        if (!first) {
            // eight counters per XCD, each dealing every eighth run of the XCD's entries to the wavefronts whose
            // number ends in c: returning atomics on one address take 11 ns each, and with one counter per XCD the
            // 1 500 tickets of a plant's batch were 16 us of them in a row
            // (fewer than eight wavefronts per XCD -- 8 walker blocks, a test's setting -- share as many counters as
            // there are of them: with eight, the runs of the counters nobody holds were dealt to nobody, and a live list
            // of more than 512 bricks kept labels no view had been applied to -- found by the fuzz sweep's big grids)
            const uint32_t nc = min(8u, per_xcd);
            const uint32_t c = t % nc;
            uint32_t n = 0;
            if (lane == 0) n = atomicAdd(&ctl->xcd_next[xcd * 8u + c].n, 1u);
            n = __builtin_amdgcn_readfirstlane(n);
            t = per_xcd + ((n / kXcdRun) * nc + c) * kXcdRun + (n % kXcdRun);
        }
        first = false;
        t = __builtin_amdgcn_readfirstlane(t);
        const uint32_t entry = ((t / kXcdRun) * 8u + xcd) * kXcdRun + (t % kXcdRun);
        if ((t / kXcdRun) * 8u * kXcdRun >= nlive) break;  // past the last run for every XCD
        if (entry >= nlive) continue;
        const uint32_t lb = __builtin_amdgcn_readfirstlane(live[entry]);
        const uint32_t il = lb / per_plane;
        const uint32_t rem = lb - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        uint32_t culled = 0;
#ifdef SC_TRACE_DENSE
        const uint64_t trb = wall_clock64();
        tr_tick += (uint32_t)(trb - tra);
        ++tr_bricks;
#endif
        // (a wavefront whose last 8 bricks had no unit to cull -- masks without structure -- asks only about every
        // eighth brick from then on: the verdicts cost a tenth of the projections they cannot spare there)
        const bool ask = nverd > 0 && (misses < 8u || (turn & 7u) == 0u);
        ++turn;
        if (ask) {  // wave-uniform
            const uint32_t u = lane >> 4, vq = lane & 15u;
            uint32_t v = 0u;
            if ((int)vq < nverd) {
                const ViewDesc d = views[vq];  // one descriptor per lane
                const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
                v = rect_verdict_cells(d, g, x, (int)(by * kBrickY), (int)(by * kBrickY) + kBrickY - 1,
                                       (int)(bz * kBrickZ + u * 16u), (int)(bz * kBrickZ + u * 16u) + 15);
            }
            const unsigned long long e = __ballot(v == 1u);  // some view carves the whole unit
            culled = ((e & 0xffffull) ? 1u : 0u) | (((e >> 16) & 0xffffull) ? 2u : 0u) |
                     (((e >> 32) & 0xffffull) ? 4u : 0u) | ((e >> 48) ? 8u : 0u);
            misses = culled ? 0u : misses + 1u;
        }
#ifdef SC_TRACE_DENSE
        const uint64_t trc = wall_clock64();
        tr_verd += (uint32_t)(trc - trb);
#endif
        // lane = column * 4 + group of 4 voxels: a square patch of the plane, the UNIT the bulk list speaks of (see Append)
        const uint32_t j = by * kBrickY + (lane >> 2);
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t k0 = bz * kBrickZ + u * 16u + (lane & 3u) * 4u;
            if ((culled >> u) & 1u) {
                if (j < g.ny && k0 < g.nz)
                    *reinterpret_cast<int4 *>(labels + ((uint64_t)il * g.ny + j) * g.nzp + k0) = make_int4(-1, -1, -1, -1);
                continue;
            }
            brick_voxels<FRESH, ALL_SAFE>(labels, g, views, nviews, init, ap, il, j, k0, lb, lane, u, nextra);
#ifdef SC_TRACE_DENSE
            ++tr_units;
#endif
        }
#ifdef SC_TRACE_DENSE
        tr_unit += (uint32_t)(wall_clock64() - trc);
#endif
    }
#ifdef SC_TRACE_DENSE
    if (lane == 0) {
        const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
        if (w < 8192u) {
            uint32_t *o = g_dense_trace + w * 8u;
            o[0] = (uint32_t)tr0; o[1] = (uint32_t)wall_clock64(); o[2] = tr_bricks; o[3] = tr_units;
            o[4] = tr_verd; o[5] = tr_unit; o[6] = tr_tick; o[7] = 0;
        }
    }
#endif
}

// The dense kernel of a launch WITHOUT survivor stages (fewer than 6 views; a single view in the
// reference's cadence, cl.py:223-226): walkers on the live list as above, and persistent FILLERS on
// the fill list the flags kernel wrote (settled bricks that are not dead yet) instead of one store
// block per strip of the grid -- after the first views nearly every brick is dead and a launch costs
// what its few live and newly settled bricks cost, not a pass over the grid.
template <bool FRESH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(102))) void carve_brick_light_kernel(
    int32_t *__restrict__ labels, GridDesc g, const ViewDesc *__restrict__ views, int nviews, int32_t init,
    uint32_t bricks_y, uint32_t bricks_z, const uint32_t *__restrict__ live, const uint32_t *__restrict__ fill_list,
    const ListCtl *ctl, uint32_t nwalkers, uint32_t parity) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t per_plane = bricks_y * bricks_z;
    if (blockIdx.x >= nwalkers) {
        const uint32_t nfill = ctl->nfill[parity], nfillers = gridDim.x - nwalkers;
        const bool vec = (g.nzp & 3u) == 0;
        const int32_t kept = init == 0 ? 1 : init;
        // 64 entries per load (one per lane), handed out with v_readlane: one round trip per 64 bricks
        for (uint32_t base = (blockIdx.x - nwalkers) * 64u; base < nfill; base += nfillers * 64u) {
            const uint32_t mine = (base + lane < nfill) ? fill_list[base + lane] : 0u;
            const uint32_t n = min(64u, nfill - base);
            for (uint32_t q = 0; q < n; ++q) {
                const uint32_t ent = __builtin_amdgcn_readlane(mine, q);
                const bool isfull = (ent >> 31) != 0, isunt = ((ent >> 30) & 1u) != 0;
                if (isunt && !FRESH) continue;  // kept and unseen: the labels stay
                const uint32_t lb = ent & 0x3fffffffu;
                const uint32_t il = lb / per_plane;
                const uint32_t rem = lb - il * per_plane;
                const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
                const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4), k0 = bz * kBrickZ + (lane & 15) * 4;
                if (j >= g.ny || k0 >= g.nz) continue;
                int32_t *p = labels + ((uint64_t)il * g.ny + j) * g.nzp + k0;
                const uint32_t nv4 = min(4u, g.nz - k0);
                if (!isfull || FRESH) {
                    const int32_t val = isunt ? init : (isfull ? kept : -1);
                    if (vec) {
                        typedef int v4i __attribute__((ext_vector_type(4)));
                        v4i vv = {val, val, val, val};
                        __builtin_nontemporal_store(vv, reinterpret_cast<v4i *>(p));
                    } else {
                        for (uint32_t e = 0; e < nv4; ++e) p[e] = val;
                    }
                } else {  // kept whole: 0 -> 1, the rest as it is (backprojection.c:81)
                    for (uint32_t e = 0; e < nv4; ++e)
                        if (p[e] == 0) p[e] = 1;
                }
            }
        }
        return;
    }
    const uint32_t nlive = ctl->nlive[parity];
    const Append none{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3, per_xcd = nwalkers >> 3;
    for (uint32_t t = seq; ; t += per_xcd) {
        const uint32_t entry = ((t / kXcdRun) * 8u + xcd) * kXcdRun + (t % kXcdRun);
        if ((t / kXcdRun) * 8u * kXcdRun >= nlive) break;  // past the last run for every XCD
        if (entry >= nlive) continue;
        const uint32_t lb = live[entry];
        const uint32_t il = lb / per_plane;
        const uint32_t rem = lb - il * per_plane;
        const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
        const uint32_t j = by * kBrickY + wave * 4 + (lane >> 4);
        brick_voxels<FRESH>(labels, g, views, nviews, init, none, il, j, bz * kBrickZ + (lane & 15) * 4, lb, lane);
    }
}

// One view per launch (the reference's schedule, cl.py:223-226): the descriptor travels in
// the kernel arguments (no copy, no host-side wait), and each lane walks kStreamGroups
// 16-byte groups with the next group's state load already in flight -- after the first view
// nearly every wavefront only streams its state through and leaves.
template <bool FRESH, bool VEC>
__global__ __launch_bounds__(kBlock) void carve_kernel_1(int32_t *__restrict__ labels, GridDesc g,
                                                         ViewDesc view, int32_t init) {
    constexpr int G = (!FRESH && VEC) ? kStreamGroups : 1;
    uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    uint64_t grp = (uint64_t)lb * (kBlock * G) + threadIdx.x;
    Append none{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
    int4 cur = make_int4(-1, -1, -1, -1);
    // streaming loads: the state (512 MiB) is far bigger than the Infinity Cache, every view
    // reads all of it once
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto stream_load = [&](uint64_t gidx) {
        v4i q = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(labels + gidx * 4));
        return make_int4(q.x, q.y, q.z, q.w);
    };
    if (!FRESH && VEC && grp < g.ngroups) cur = stream_load(grp);  // carve_group takes the stored labels from `cur`
#pragma unroll 1
    for (int s = 0; s < G; ++s, grp += kBlock) {
        int4 nxt = make_int4(-1, -1, -1, -1);
        if (G > 1 && s + 1 < G && grp + kBlock < g.ngroups)
            nxt = stream_load(grp + kBlock);
        if (grp < g.ngroups) carve_group<FRESH, VEC>(labels, g, &view, 1, init, grp, cur, none);
        cur = nxt;
    }
}
