// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- the survivor stages (carve_list_kernel), the bulk units' verdicts and the special kernel (bulk units, late bricks, the dense fallback).

// Work items of the bulk units (see unit_verdicts): (half a unit = 8 columns x 16 voxels, up to 16 of the
// views that have to project its voxels, as a mask over 64 consecutive views).  The final list stage's
// wavefronts take them after their own spans; items of one unit may run side by side, which is exact for the
// same reason as the spans of one chunk: -1 is a plain store, 0 -> 1 a compare-and-swap on 0.
struct UnitItems {
    const uint4 *items;       // null: none.  .x = unit * 2 + half, .y = first view of the mask, .z / .w = the mask
    uint32_t cap;             // items per sub-list (counts in ctl->count[4])
    const ViewDesc *views;    // every view of the batch (the items' view numbers index this)
    uint32_t bricks_y, bricks_z;
};

// Bulk units of a batch that has too few of them for their verdicts to be asked (fewer than `floor`, see
// carve_special_kernel): the FIRST survivor stage takes them as they are -- a unit's 256 voxels as chunks of its
// own, the labels read where a list chunk reads its entries -- and appends what is left alive to its output list
// like any other survivor.  Nobody copies them anywhere in between.
// Room for `total` more entries in a sub-list, or 0xffffffff (see ListCounter: the first reservation that fails
// marks where the written entries end; the dense stage's own appends raise the overflow flag instead, which makes
// every reader ignore the lists altogether).
__device__ __forceinline__ uint32_t list_reserve(ListCounter *counter, uint32_t total, uint32_t cap, uint32_t lane) {
    uint32_t base = 0;
    if (lane == 0) {
        base = atomicAdd(&counter->n, total);
        if (base > cap || total > cap - base) {
            atomicMax(&counter->cut, ~min(base, cap));
            base = 0xffffffffu;
        }
    }
    return __shfl(base, 0);
}

struct UnitSpill {
    const uint32_t *units;  // null: no bulk list.  [kSub][cap] unit ids, counts in ctl->count[3]
    uint32_t cap, floor;
    uint32_t bricks_y, bricks_z;
};

// Fused carve, sparse phase: one lane per SURVIVOR.  Reads the survivor sub-lists a previous
// stage appended and applies views with every lane busy, two views per iteration (two
// independent projection chains and two gathers in flight per lane).  A persistent grid of
// wavefronts walks the work items; the counts live in device memory, so the host never waits
// to learn how many survivors there are.
//   FINAL == false: an item is a 64-entry chunk and ALL `nviews` views; labels that change
//     are written (carved -> -1 at once, 0 -> 1 at the end) and what is still alive is appended
//     to `lout` for the next stage.
//   FINAL == true : an item is a 64-entry chunk times a GROUP of `vgsize` views, so that there
//     are many more items than wavefronts (no tail); groups of one chunk may run concurrently
//     on different wavefronts, which is exact because a carve is a plain store of -1 (final,
//     idempotent) and a 0 -> 1 promotion is a compare-and-swap on 0 (it can never undo a -1).
// (at most 96 SGPRs -- the eight lane masks of four views a turn need them: with 82-96 the CU admits 7 such blocks
// instead of 8, with 98+ only 6 -- MI355X_MICROARCH.md, "Residency"; the list stages run 6 blocks per CU, store blocks
// included, and the store blocks need the slots the list blocks leave)
template <bool FINAL, int P, bool ALL_SAFE = false>  // P voxels per lane (an item is a chunk of 64 * P entries); ALL_SAFE: see project()
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(96))) void carve_list_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                            const ViewDesc *__restrict__ views,
                                                            int nviews,
                                                            const uint32_t *__restrict__ lin,
                                                            uint32_t *__restrict__ lout,
                                                            ListCtl *ctl, int sin, int sout,
                                                            uint32_t subcap, int vgsize, CullStores cs, UnitItems ui,
                                                            UnitSpill us, int nrest) {  // nrest: views from `views` to the batch's last
    __shared__ uint32_t pref[kSub + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t bx = blockIdx.x, gdim = gridDim.x;
    // A list stage with deferred stores has STORE blocks behind its persistent list blocks: the stage's
    // own work is projection arithmetic, the -1 fill of the bricks the flags kernel found empty is HBM
    // writes, so the two run side by side instead of one after the other.  (Since round 4 cut the
    // arithmetic the final stage of a plant is bound by the fill beside it -- about 5.3 TB/s there,
    // whatever the store pattern: DESIGN_APPENDIX.md 12.)
    const bool split = cs.flags != nullptr;
    const uint32_t nown = cs.nstrips - cs.first, nvirt = nown + cs.spec;  // the share proper, and the prefilled strips behind it
    const uint32_t nstore = split ? (cs.fill_blocks ? cs.fill_blocks : nvirt) : 0u;
    const uint32_t nbid = gdim - nstore;
    if (split && bx >= nbid) {
        // one short block per strip, or (fill_blocks > 0) that many blocks walking the strips: a
        // wavefront's stores do not hold it up, so few of them keep the write path busy and the
        // wavefront slots go to the list blocks
        constexpr uint32_t kAhead = 4;  // strips whose flags are fetched together
        for (uint32_t t = bx - nbid; t < nvirt; t += nstore * kAhead) {
            uint32_t f[kAhead];
#pragma unroll
            for (uint32_t u = 0; u < kAhead; ++u) {
                const uint32_t tt = t + u * nstore;
                f[u] = tt < nvirt ? strip_flag(cs.flags, tt < nown ? cs.first + tt : tt - nown, cs.bricks_z) : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kAhead; ++u) {
                const uint32_t tt = t + u * nstore;
                if (tt < nvirt)
                    store_culled_bricks(labels, g, f[u], tt < nown ? cs.first + tt : tt - nown, cs.bricks_y, cs.bricks_z,
                                        Fill{cs.kept, cs.fresh, cs.init}, tt >= nown);
            }
        }
        return;
    }
    if (ctl->overflow) return;  // the special kernel's dense pass has applied the remaining views instead
    const uint32_t bid = bx;
    constexpr uint32_t CH = 64u * P;
    {
        uint32_t c = (list_count(ctl->count[sin][tid], subcap) + CH - 1u) / CH;  // kSub == kBlock
        if (tid == 0) pref[0] = 0;
        pref[tid + 1] = c;
        __syncthreads();
        for (uint32_t off = 1; off < kSub; off <<= 1) {
            uint32_t val = pref[tid + 1];
            uint32_t add = (tid >= off) ? pref[tid + 1 - off] : 0u;
            __syncthreads();
            pref[tid + 1] = val + add;
            __syncthreads();
        }
    }
    const uint32_t chunks = pref[kSub];
    const uint32_t lane = tid & 63u;
    __shared__ uint32_t ipref[kSub + 1];  // FINAL: the work items' prefix; else: the spilled bulk units'
    const bool with_items = FINAL && ui.items != nullptr;  // grid-uniform
    uint32_t uchunks = 0;  // chunks made of spilled bulk units, behind the list's own
    constexpr uint32_t kUnitParts = 256u / (64u * P);  // chunks per unit
    if (!FINAL && us.units != nullptr) {  // grid-uniform
        const uint32_t c = min(ctl->count[3][tid].n, us.cap);
        if (tid == 0) ipref[0] = 0;
        ipref[tid + 1] = c;
        __syncthreads();
        for (uint32_t off = 1; off < kSub; off <<= 1) {
            const uint32_t val = ipref[tid + 1];
            const uint32_t add = (tid >= off) ? ipref[tid + 1 - off] : 0u;
            __syncthreads();
            ipref[tid + 1] = val + add;
            __syncthreads();
        }
        if (ipref[kSub] < us.floor) uchunks = ipref[kSub] * kUnitParts;  // (else the special kernel has dealt with them)
    }
    if (with_items) {
        const uint32_t c = list_count(ctl->count[4][tid], ui.cap);
        if (tid == 0) ipref[0] = 0;
        ipref[tid + 1] = c;
        __syncthreads();
        for (uint32_t off = 1; off < kSub; off <<= 1) {
            const uint32_t val = ipref[tid + 1];
            const uint32_t add = (tid >= off) ? ipref[tid + 1 - off] : 0u;
            __syncthreads();
            ipref[tid + 1] = val + add;
            __syncthreads();
        }
    }
    const uint64_t nworkers = (uint64_t)nbid * (kBlock / 64);
    // the wavefront index must be a scalar for the compiler, or everything derived from the
    // item (view range, descriptors) is treated as divergent and fetched with vector loads
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // FINAL: the (chunk, view) pairs, chunk-major, are cut into one SPAN per wavefront -- every
    // wavefront gets the same number of projections whatever the counts are (with whole
    // (chunk, view group) items 17 k items over 4096 wavefronts meant 5 items for some and 4 for
    // others), and a span crosses a chunk boundary once or twice, so the decode of the entries is
    // paid once or twice per wavefront.  `vgsize` only rounds the span length.
    // Not FINAL: an item is a chunk and all the views, dealt round-robin.
    const uint64_t total = FINAL ? (uint64_t)chunks * (uint32_t)nviews : (uint64_t)chunks + uchunks;
    uint64_t per = 1;
    if (FINAL) {
        per = (total + nworkers - 1) / nworkers;
        const uint64_t r = (uint64_t)max(vgsize, 1);
        per = (per + r - 1) / r * r;
    }
    const uint64_t wid = (uint64_t)bid * (kBlock / 64) + wave;
    uint64_t pos = FINAL ? min(total, wid * per) : wid;
    const uint64_t end = FINAL ? min(total, pos + per) : total;
    while (pos < end) {
        uint32_t c;
        int v0, v1;
        if (FINAL) {
            c = (uint32_t)(pos / (uint32_t)nviews);
            v0 = (int)(pos - (uint64_t)c * (uint32_t)nviews);
            v1 = (int)min((uint64_t)nviews, (uint64_t)v0 + (end - pos));
            pos += (uint64_t)(v1 - v0);
        } else {
            c = (uint32_t)pos;
            v0 = 0;
            v1 = nviews;
            pos += nworkers;
        }
        const bool from_unit = !FINAL && c >= chunks;  // wave-uniform: a chunk of a spilled bulk unit
        const uint32_t *pf = from_unit ? ipref : pref;
        const uint32_t key = from_unit ? (c - chunks) / kUnitParts : c;
        uint32_t lo = 0, hi = kSub;  // largest s with pf[s] <= key (wave-uniform)
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (pf[mid] <= key) lo = mid; else hi = mid;
        }
        const uint32_t s = lo;
        const uint32_t cnt = list_count(ctl->count[sin][s], subcap);
        // P voxels per lane: the descriptor traffic and the scalar bookkeeping of a view are shared,
        // and a lane has P * U independent projection chains and gathers in flight.
        // What a lane knows about its voxels -- alive, label still 0, 0 -> 1 pending -- is kept as WAVEFRONT masks in
        // scalar registers (round 4): the comparisons of a projection leave lane masks anyway, so the whole of
        // backprojection.c:79-83 becomes scalar and/or on them, and vector instructions are spent on a store or a
        // compare-and-swap only in the turns where some lane has one to make (the byte-per-lane booleans the compiler
        // made of `bool alive[P]` cost 8 vector instructions per voxel and turn).
        uint32_t idx[P];
        unsigned long long alive_m[P], zero_m[P], flip_m[P];
        float x[P], y[P], z[P];
        uint32_t unit = 0, upart = 0;
        if (from_unit) {
            unit = __builtin_amdgcn_readfirstlane(us.units[(size_t)s * us.cap + (key - ipref[s])]);
            upart = (c - chunks) % kUnitParts;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (from_unit) {
                // voxel v of the unit (16 columns x 16 voxels of brick unit >> 2, its z quarter unit & 3): column v >> 4
                const uint32_t v = upart * CH + (uint32_t)p * 64u + lane;
                const uint32_t lb = unit >> 2, per_plane = us.bricks_y * us.bricks_z;
                const uint32_t il = lb / per_plane, rem = lb - il * per_plane;
                const uint32_t by = rem / us.bricks_z, bz = rem - by * us.bricks_z;
                const uint32_t j = by * kBrickY + (v >> 4), k = bz * kBrickZ + (unit & 3u) * 16u + (v & 15u);
                const bool inside = j < g.ny && k < g.nz;
                idx[p] = inside ? (il * g.ny + j) * g.nzp + k : 0u;
                int32_t lab = -1;
                if (inside) lab = labels[idx[p]];
                alive_m[p] = __ballot(lab != -1);   // backprojection.c:67
                zero_m[p] = __ballot(lab == 0);
            } else {
                const uint32_t e = (c - pref[s]) * CH + (uint32_t)p * 64u + lane;
                const bool in = e < cnt;
                uint32_t entry = 0;
                if (in) entry = lin[(size_t)s * subcap + e];
                idx[p] = entry & 0x7fffffffu;
                alive_m[p] = __ballot(in);
                zero_m[p] = __ballot((entry >> 31) != 0);  // label is still 0
            }
            flip_m[p] = 0;
            const uint32_t col = fdiv(idx[p], g.by_nzp);  // entries index the padded rows (below 2^31)
            const uint32_t k = idx[p] - col * g.nzp;
            const uint32_t il = fdiv(col, g.by_ny);
            const uint32_t j = col - il * g.ny;
            x[p] = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // backprojection.c:71-73
            y[p] = g.oy + (float)(int)j * g.vs;
            z[p] = g.oz + (float)(int)k * g.vs;
        }
        // U views per iteration: four (8 projection chains and 8 gathers in flight per lane at P = 2).  The final stage ran
        // two until round 4's end -- it was bound by its arithmetic then; beside the fill that bounds it now, four took
        // it from 67.2 to 63.3 us and the batch from 0.160 to 0.1545 ms (same box, four runs each).
        constexpr int U = P >= 4 ? 1 : 4;
        for (;;) {  // (once; a second time over the views behind this stage's for a chunk whose survivors find no room)
        for (int vi = v0; vi < v1; vi += U) {
            unsigned long long any = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) any |= alive_m[p];
            if (any == 0) break;
            unsigned long long okm[U][P];
            uint32_t w[U][P];
            int sh[U][P];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                // past the end of the range the last view is applied once more: a view applied twice
                // changes nothing (a carve is final, a kept 0 is already 1), and nothing per lane has
                // to know whether the slot was real
                const ViewDesc d = views[vi + q < v1 ? vi + q : vi];
                // every field in scalar registers NOW: left alone the compiler fetches Wf/Hf,
                // tiles_x and the mask pointer one by one where they are first used, three
                // more scalar-load round trips inside each projection
                asm volatile("" ::"s"(d.Wf), "s"(d.Hf), "s"(d.strip), "s"(d.mask));
                const uint32_t tile_row = (uint32_t)d.strip;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    int vv;
                    // dead lanes project along (their carve below is masked): cheaper than a per-lane test here
                    const bool ok = project<ALL_SAFE>(d.R[0] * x[p] + d.R[1] * y[p], d.R[3] * x[p] + d.R[4] * y[p],
                                                      d.R[6] * x[p] + d.R[7] * y[p], z[p], d, sh[q][p], vv, okm[q][p]);
                    w[q][p] = 0;
                    if (ok) w[q][p] = load_mask_at(d.mask, mask_byte_offset(sh[q][p], vv, tile_row));
                }
            }
            // U applications of backprojection.c:79-83; a zero pixel in any of the views wins
#pragma unroll
            for (int p = 0; p < P; ++p) {
                unsigned long long carve = 0, keep = 0;
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const unsigned long long fg = __ballot(__builtin_amdgcn_ubfe(w[q][p], (uint32_t)sh[q][p], 1u) != 0u);  // (a word not loaded is 0)
                    carve |= okm[q][p] & ~fg;
                    keep |= fg;
                }
                carve &= alive_m[p];
                if (carve != 0) {  // wave-uniform
                    if (__builtin_amdgcn_inverse_ballot_w64(carve)) labels[idx[p]] = -1;
                    alive_m[p] &= ~carve;
                    zero_m[p] &= ~carve;
                }
                keep &= zero_m[p];
                if (keep != 0) {
                    if (FINAL) {
                        if (__builtin_amdgcn_inverse_ballot_w64(keep)) atomicCAS(&labels[idx[p]], 0, 1);
                    } else {
                        flip_m[p] |= keep;
                    }
                    zero_m[p] &= ~keep;
                }
            }
        }
        if (FINAL) break;
        bool noroom = false;
        if (v1 < nrest) {  // views behind this stage: what is still alive goes on the output list
            uint32_t tot = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) tot += (uint32_t)__popcll(alive_m[p]);
            if (tot != 0) {  // wave-uniform
                // The survivors of sub-list s do not outnumber its entries, but spilled bulk units append here too:
                // without room (the counter's `cut` marks where the written entries end) this wavefront takes its chunk through the
                // remaining views itself -- these voxels are on no list, nobody else touches them
                uint32_t base = list_reserve(&ctl->count[sout][s], tot, subcap, lane);
                noroom = base == 0xffffffffu;
                if (!noroom) {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        if (__builtin_amdgcn_inverse_ballot_w64(alive_m[p]))
                            lout[(size_t)s * subcap + base + lanes_below(alive_m[p])] = idx[p] | (__builtin_amdgcn_inverse_ballot_w64(zero_m[p]) ? 0x80000000u : 0u);
                        base += (uint32_t)__popcll(alive_m[p]);
                    }
                }
            }
        }
        if (!noroom) break;
        v0 = v1;
        v1 = nrest;
        }
        if (!FINAL) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const unsigned long long m = alive_m[p] & flip_m[p];
                if (m != 0 && __builtin_amdgcn_inverse_ballot_w64(m)) labels[idx[p]] = 1;
            }
        }
    }
    if (with_items) {
        // the bulk units' work items, dealt round-robin: two voxels per lane, the views the item names, two
        // per turn as above
        const uint32_t itotal = ipref[kSub];
        const uint32_t per_plane = ui.bricks_y * ui.bricks_z;
        for (uint32_t i = (uint32_t)wid; i < itotal; i += (uint32_t)nworkers) {
            uint32_t lo = 0, hi = kSub;  // largest s with ipref[s] <= i (wave-uniform)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ipref[mid] <= i) lo = mid; else hi = mid;
            }
            uint4 it = ui.items[(size_t)lo * ui.cap + (i - ipref[lo])];
            it.x = __builtin_amdgcn_readfirstlane(it.x);
            it.y = __builtin_amdgcn_readfirstlane(it.y);
            it.z = __builtin_amdgcn_readfirstlane(it.z);
            it.w = __builtin_amdgcn_readfirstlane(it.w);
            const uint32_t unit = it.x >> 1, lb = unit >> 2;
            const uint32_t il = lb / per_plane, rem = lb - il * per_plane;
            const uint32_t by = rem / ui.bricks_z, bz = rem - by * ui.bricks_z;
            // half h of a unit: its columns 8 h .. 8 h + 7; lane = column * 8 + (voxel & 7), p = voxel >> 3: a lane's two
            // voxels share their column, so the x and y terms of a view's three sums are computed once for both
            const uint32_t j = by * kBrickY + (it.x & 1u) * 8u + (lane >> 3), k0 = bz * kBrickZ + (unit & 3u) * 16u + (lane & 7u);
            const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // backprojection.c:71-73
            const float y = g.oy + (float)(int)j * g.vs;
            uint32_t idx[2];
            unsigned long long alive_m[2], zero_m[2];
            float z[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t k = k0 + 8u * (uint32_t)p;
                const bool inside = j < g.ny && k < g.nz;
                idx[p] = (il * g.ny + j) * g.nzp + k;
                int32_t lab = -1;
                if (inside) lab = labels[idx[p]];
                alive_m[p] = __ballot(lab != -1);
                zero_m[p] = __ballot(lab == 0);
                z[p] = g.oz + (float)(int)k * g.vs;
            }
            unsigned long long m = ((unsigned long long)it.w << 32) | it.z;
            const uint32_t vbase = it.y;
            while (m != 0) {
                if ((alive_m[0] | alive_m[1]) == 0) break;
                const uint32_t a = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint32_t b = a;  // a lone view is applied twice: nothing changes the second time
                if (m != 0) {
                    b = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                }
                unsigned long long okm[2][2];
                uint32_t w[2][2];
                int sh[2][2];
                // both descriptors in scalar registers before either view projects: one scalar-load round trip per turn
                const ViewDesc dq[2] = {scalar_desc(ui.views, vbase + a), scalar_desc(ui.views, vbase + b)};  // (`ui.views`: a pointer inside a struct)
                asm volatile("" ::"s"(dq[0].Wf), "s"(dq[0].Hf), "s"(dq[0].strip), "s"(dq[0].mask), "s"(dq[0].R[0]), "s"(dq[0].K[0]), "s"(dq[0].t[0]),
                             "s"(dq[1].Wf), "s"(dq[1].Hf), "s"(dq[1].strip), "s"(dq[1].mask), "s"(dq[1].R[0]), "s"(dq[1].K[0]), "s"(dq[1].t[0]));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const ViewDesc &d = dq[q];
                    const uint32_t tile_row = (uint32_t)d.strip;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        int vv;
                        const bool ok = project<ALL_SAFE>(d.R[0] * x + d.R[1] * y, d.R[3] * x + d.R[4] * y,
                                                          d.R[6] * x + d.R[7] * y, z[p], d, sh[q][p], vv, okm[q][p]);
                        w[q][p] = 0;
                        if (ok) w[q][p] = load_mask_at(d.mask, mask_byte_offset(sh[q][p], vv, tile_row));
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    unsigned long long carve = 0, keep = 0;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned long long fg = __ballot(__builtin_amdgcn_ubfe(w[q][p], (uint32_t)sh[q][p], 1u) != 0u);
                        carve |= okm[q][p] & ~fg;
                        keep |= fg;
                    }
                    carve &= alive_m[p];
                    if (carve != 0) {
                        if (__builtin_amdgcn_inverse_ballot_w64(carve)) labels[idx[p]] = -1;
                        alive_m[p] &= ~carve;
                        zero_m[p] &= ~carve;
                    }
                    keep &= zero_m[p];
                    if (keep != 0) {
                        if (__builtin_amdgcn_inverse_ballot_w64(keep)) atomicCAS(&labels[idx[p]], 0, 1);
                        zero_m[p] &= ~keep;
                    }
                }
            }
        }
    }
}

struct LateBricks {           // FULL candidates that turned out not to be (see brick_confirm_kernel)
    const uint32_t *late;     // null: the batch had no open candidates
    uint32_t nbricks;         // the list's room: the bricks that take the units' road are entered from its far end
    const ViewDesc *allviews; // every view of the batch
    const uint8_t *flags;
    int32_t nall, init, fresh;
    uint32_t bricks_y, bricks_z;
};

// A unit of a LATE brick (a FULL candidate some later view did not keep whole after all) through every view of
// the batch, one wavefront: the views are first asked about the unit as a whole, 64 at a time, one view per
// lane, at the cell level -- a view that sees it entirely over background carves all of it, views that see it
// entirely over foreground or not at all have nothing to say about its voxels one by one -- and only the
// others project them, two per turn.  (A brick inside a solid object lies over foreground in nearly all the
// views, one at the edge of the pictures outside nearly all.)
template <bool FRESH>
__device__ __forceinline__ void late_unit(int32_t *__restrict__ labels, const GridDesc &g,
                                          const ViewDesc *__restrict__ views, int nall, int32_t init, uint32_t unit,
                                          uint32_t bricks_y, uint32_t bricks_z, uint32_t lane) {
    const uint32_t lb = unit >> 2, w = unit & 3u;
    const uint32_t per_plane = bricks_y * bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / bricks_z, bz = rem - by * bricks_z;
    const int j0 = (int)(by * kBrickY), kb = (int)(bz * kBrickZ + w * 16u);  // 16 columns x 16 voxels
    const uint32_t j = (uint32_t)j0 + (lane >> 2), k0 = (uint32_t)kb + (lane & 3u) * 4u;
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    int32_t *p = labels + ((uint64_t)il * g.ny + j) * g.nzp + k0;  // the pitch is a multiple of 64: 16-byte groups
    int32_t lab[4] = {-1, -1, -1, -1}, was[4];  // what a lane does not own counts as carved
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nvalid) lab[e] = init;
    } else if (inside) {
        const int4 q = *reinterpret_cast<const int4 *>(p);
        lab[0] = q.x; lab[1] = q.y; lab[2] = q.z; lab[3] = q.w;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e >= nvalid) lab[e] = -1;  // row padding behind the last voxel
    }
    unsigned long long alive[4], kept[4];  // see two_views
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
    bool seen = false;
    for (int base = 0; base < nall; base += 64) {
        if ((alive[0] | alive[1] | alive[2] | alive[3]) == 0) break;
        const int vi = base + (int)lane;
        uint32_t v = 8u;  // no such view
        if (vi < nall) {
            const ViewDesc d = views[vi];  // one descriptor per lane
            v = d.cmask != nullptr ? rect_verdict_cells(d, g, x, j0, j0 + kBrickY - 1, kb, kb + 15) : 0u;
        }
        const unsigned long long empty = __ballot(v == 1u), full = __ballot(v == 2u);
        unsigned long long need = __ballot(v == 0u);
        if (empty != 0) {  // some view carves every voxel of the unit
#pragma unroll
            for (int e = 0; e < 4; ++e) alive[e] = 0;
            break;
        }
        seen |= full != 0;
        while (need != 0) {
            if ((alive[0] | alive[1] | alive[2] | alive[3]) == 0) break;
            const int a = __builtin_ctzll(need);
            need &= need - 1;
            int b = a;
            const bool two = need != 0;
            if (two) {
                b = __builtin_ctzll(need);
                need &= need - 1;
            }
            const ViewDesc da = scalar_desc(views, (uint32_t)(base + a));
            const ViewDesc db = scalar_desc(views, (uint32_t)(base + b));
            two_views(da, db, two, x, y, z, alive, kept);
        }
    }
    if (seen) {  // :81 by a view that kept the whole unit
#pragma unroll
        for (int e = 0; e < 4; ++e) kept[e] = ~0ull;
    }
    labels_behind(lab, alive, kept);
    const bool changed = FRESH || lab[0] != was[0] || lab[1] != was[1] || lab[2] != was[2] || lab[3] != was[3];
    if (inside && changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
}

// The UNITS of the bulk list (a wavefront's share of a live brick -- its 16 columns, voxels 16w .. 16w + 15 of
// each: a square patch of the plane -- with most of its voxels alive after the dense views) are asked about as a whole before anything projects
// their voxels: every remaining view at once, one view per lane, at the cell level (rect_verdict_cells).
//   some view sees the unit entirely over background (EMPTY): every voxel is carved, done;
//   views that see it entirely over foreground (FULL) make a label 0 a 1 (backprojection.c:81) here and now,
//   and like the views that do not see it at all (OUTSIDE) have nothing more to say;
//   the UNDECIDED views are the only ones that have to project its voxels: they become work items
//   (half a unit x up to 16 of those views, see UnitItems) for the final list stage -- or, when that would
//   be no cheaper than the ordinary survivor lists (few voxels alive, most views undecided), the unit's
//   voxels are appended to the first list like any other survivor.
struct UnitJob {
    const uint32_t *units;    // null: no bulk list.  [kSub][cap] unit ids (brick * 4 + wavefront), counts in ctl->count[3]
    uint32_t cap;
    uint4 *items;             // [kSub][icap] work items out, counts in ctl->count[4]
    uint32_t icap;
    const ViewDesc *views;    // every view of the batch
    int32_t nall, ndense;     // ... their number (<= 128), and how many of them the dense stage has applied
    uint32_t bricks_y, bricks_z;
    int32_t *labels;
    uint32_t *list;           // the first survivor list and the room of its sub-lists (counts in ctl->count[0])
    uint32_t subcap;
    uint32_t bias;            // items are chosen when their turns * 16 <= bias * the turns the lists would take
    uint32_t floor;           // fewer bulk units than this in the whole batch: no verdicts, the first survivor stage
                              // takes their voxels as they are (see carve_special_kernel, UnitSpill)
};

// (second: the verdicts of views 64 .. nall - 1 about this unit, asked beforehand for several units in one round --
// see unit_second_word; bit k speaks of view 64 + k)
struct SecondWord {
    unsigned long long empty, full, need;
};
// (first: the lane's own view's descriptor, view `lane` of the batch -- the same for every unit a wavefront asks about,
// so the caller loads it once; the special kernel has the registers: 94 of the 128 that four wavefronts per SIMD allow)
__device__ __forceinline__ void unit_verdicts(const UnitJob &uj, const GridDesc &g, ListCtl *ctl, uint32_t unit,
                                              uint32_t sub, uint32_t lane, const SecondWord &second, const ViewDesc &first) {
    const uint32_t lb = unit >> 2, w = unit & 3u;
    const uint32_t per_plane = uj.bricks_y * uj.bricks_z;
    const uint32_t il = lb / per_plane;
    const uint32_t rem = lb - il * per_plane;
    const uint32_t by = rem / uj.bricks_z, bz = rem - by * uj.bricks_z;
    const int j0 = (int)(by * kBrickY), kb = (int)(bz * kBrickZ + w * 16u);  // 16 columns x 16 voxels
    const uint32_t j = (uint32_t)j0 + (lane >> 2), k0 = (uint32_t)kb + (lane & 3u) * 4u;
    const bool inside = j < g.ny && k0 < g.nz;
    const int nvalid = inside ? (int)min(4u, g.nz - k0) : 0;
    const uint32_t elem = (il * g.ny + j) * g.nzp + k0;
    int32_t *p = uj.labels + elem;  // the pitch is a multiple of 64: 16-byte groups
    // (the labels are asked for here and looked at behind the verdict round: one memory round trip, not two in a row)
    int4 q = make_int4(-1, -1, -1, -1);  // what a lane does not own counts as carved
    if (inside) q = *reinterpret_cast<const int4 *>(p);
    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;  // :71, global plane index
    unsigned long long need[2] = {0ull, second.need};
    bool seen = second.full != 0, empty = second.empty != 0;
    {
        const int vi = (int)lane;  // the first 64 views, one per lane
        uint32_t v = 8u;  // no such view, or one the dense stage has applied
        if (vi < uj.nall && vi >= uj.ndense)
            v = first.cmask != nullptr ? rect_verdict_cells(first, g, x, j0, j0 + kBrickY - 1, kb, kb + 15) : 0u;
        empty |= __ballot(v == 1u) != 0;
        seen |= __ballot(v == 2u) != 0;
        need[0] = __ballot(v == 0u);
    }
    int32_t lab[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (e >= nvalid) lab[e] = -1;  // row padding behind the last voxel
    uint32_t alive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (lab[e] != -1) alive |= 1u << e;  // :67
    unsigned long long b[4];
    uint32_t nalive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        b[e] = __ballot((alive >> e) & 1u);
        nalive += (uint32_t)__popcll(b[e]);
    }
    // turns of (128 voxels x 2 views) the unit's voxels would take in the survivor lists
    const uint32_t list_cost = ((nalive + 127u) >> 7) * (((uint32_t)(uj.nall - uj.ndense) + 1u) >> 1);
    if (empty) {  // some view carves every voxel of the unit
        if (inside && alive != 0) *reinterpret_cast<int4 *>(p) = make_int4(-1, -1, -1, -1);
        return;
    }
    if (seen) {
        bool changed = false;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (lab[e] == 0) { lab[e] = 1; changed = true; }  // :81 by a view that keeps the whole unit
        if (inside && changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    }
    const uint32_t nneed = (uint32_t)__popcll(need[0]) + (uint32_t)__popcll(need[1]);
    const unsigned long long anyalive = __ballot(alive != 0);
    if (anyalive == 0 || nneed == 0) return;  // wave-uniform: the labels are final
    // the undecided views of each 64-view word in pieces of up to 16; one item per (half with something
    // alive, word, piece): lane = piece * 4 + word * 2 + half
    unsigned long long pm[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bool bit = (need[h] >> lane) & 1ull;
        const uint32_t piece = lanes_below(need[h]) >> 4;  // this lane's view is the (16 piece + ..)-th undecided one
#pragma unroll
        for (int q = 0; q < 4; ++q) pm[h][q] = __ballot(bit && piece == (uint32_t)q);
    }
    const uint32_t hq = lane & 1u, wq = (lane >> 1) & 1u, pq = lane >> 2;
    unsigned long long mymask = 0ull;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (wq == (uint32_t)h && pq == (uint32_t)q) mymask = pm[h][q];
    const uint32_t halves = ((uint32_t)(anyalive & 0xffffffffull) != 0u ? 1u : 0u) + ((uint32_t)(anyalive >> 32) != 0u ? 1u : 0u);
    const bool half_alive = ((anyalive >> (32u * hq)) & 0xffffffffull) != 0;
    const bool mine = lane < 16u && mymask != 0ull && half_alive;
    const unsigned long long im = __ballot(mine);
    const uint32_t nitems = (uint32_t)__popcll(im);
    // turns of (128 voxels x 2 views): the items' against what the unit's voxels would take in the lists
    const uint32_t item_cost = halves * ((nneed + 1u) / 2u) + nitems;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (item_cost * 16u <= uj.bias * list_cost) {
        const uint32_t pos = list_reserve(&ctl->count[4][sub], nitems, uj.icap, lane);
        if (pos != 0xffffffffu) {
            if (mine)
                uj.items[(size_t)sub * uj.icap + pos + (uint32_t)__popcll(im & below)] =
                    make_uint4(unit * 2u + hq, wq * 64u, (uint32_t)mymask, (uint32_t)(mymask >> 32));
            return;
        }
        // (no room among the items -- the counter's `cut` marks where the written ones end: the voxels take the list)
    }
    const uint32_t base = list_reserve(&ctl->count[0][sub], nalive, uj.subcap, lane);
    if (base == 0xffffffffu) {
        // no room in the sub-list (its counter's `cut` marks where the written entries end): this wavefront takes the unit through
        // every view of the batch itself.  The overflow flag is the dense stage's alone -- the blocks of this
        // kernel read it when they start -- and views applied twice change nothing.
        late_unit<false>(uj.labels, g, uj.views, uj.nall, 0, unit, uj.bricks_y, uj.bricks_z, lane);
        return;
    }
    uint32_t *dst = uj.list + (size_t)sub * uj.subcap + base;
    uint32_t rank = 0;  // in address order (see brick_voxels)
#pragma unroll
    for (int e = 0; e < 4; ++e) rank += (uint32_t)__popcll(b[e] & below);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if ((alive >> e) & 1u) {
            dst[rank] = (elem + (uint32_t)e) | (lab[e] == 0 ? 0x80000000u : 0u);
            ++rank;
        }
    }
}

// The kernel between the dense stage (and the confirm kernel) and the survivor stages: everything of a batch that is
// neither a brick verdict nor a survivor list.
//   * zeroes the counters of the NEXT batch (the two control blocks alternate; nobody else touches that one now),
//     so no memset sits on the stream;
//   * the units of the bulk list get their verdicts, one wavefront per unit (unit_verdicts) -- when the batch has
//     at least `uj.floor` of them: a latency chain of ~7 us per round of wavefronts is not worth a handful of units
//     (the bench's thin plant has 3 268, a bulky object 18 000 - 56 000), which the first survivor stage then takes as they are (UnitSpill).  The
//     decision is taken on the DEVICE, here and there alike, from the count the dense stage of this very batch
//     left, so the first batch of a fresh engine runs exactly like every later one;
//   * LATE bricks (FULL candidates some later view did not keep whole after all) are carved unit by unit over every
//     view of the batch (late_unit);
//   * when a survivor sub-list overflowed in the dense stage (masks that carve little), the views it has not
//     applied are applied densely here instead; the list kernels behind see the flag and leave.
struct SpecialJob {
    UnitJob uj;            // uj.units == nullptr: no bulk list
    LateBricks lb;         // lb.late == nullptr: the batch had no open candidates
    ListCtl *next;         // nullptr: nothing to zero
    const ViewDesc *rest;  // the views the dense stage has not applied (dense fallback)
    int32_t nrest;
    const uint8_t *flags;  // brick verdicts (nullptr: a launch without bricks)
    uint32_t bricks_y, bricks_z;
};

__global__ __launch_bounds__(64 * kFlagWaves) void carve_special_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                                       ListCtl *ctl, SpecialJob sj) {
    constexpr uint32_t kThreads = 64 * kFlagWaves;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint32_t nworkers = gridDim.x * kFlagWaves;
    if (sj.next != nullptr) {
        uint32_t *z = reinterpret_cast<uint32_t *>(sj.next);
        for (uint32_t i = blockIdx.x * kThreads + tid; i < sizeof(ListCtl) / 4; i += gridDim.x * kThreads) z[i] = 0u;
    }
    const bool overflow = ctl->overflow != 0u;  // written by the dense stage only: the same for every block
    if (sj.uj.units != nullptr && !overflow) {  // grid-uniform
        __shared__ uint32_t upref[kSub + 1];
        __shared__ uint32_t s_total;
        const uint32_t mine = tid < kSub ? min(ctl->count[3][tid].n, sj.uj.cap) : 0u;
        if (tid == 0) s_total = 0u;
        __syncthreads();
        uint32_t wsum = mine;
        for (int off = 32; off > 0; off >>= 1) wsum += __shfl_xor(wsum, off);
        if (lane == 0 && wsum != 0u) atomicAdd(&s_total, wsum);
        __syncthreads();
        // Fewer bulk units than the floor: nobody asks about THEM, the first survivor stage takes their voxels as they
        // are (UnitSpill -- it applies the same rule to the same counts).  The units of the candidates that failed
        // (round 5: brick_confirm_kernel, UnitRoad) are always asked -- they are on no other list.  Grid-uniform.
        const uint32_t nb = (s_total >= sj.uj.floor) ? s_total : 0u;
        const uint32_t nl = sj.lb.late != nullptr ? 4u * ctl->nlate_units : 0u;
        const uint32_t total = nb + nl;
        if (total != 0u) {
            if (tid < kSub) upref[tid + 1] = mine;
            if (tid == 0) upref[0] = 0;
            __syncthreads();
            for (uint32_t off = 1; off < kSub; off <<= 1) {
                uint32_t val = 0, add = 0;
                if (tid < kSub) {
                    val = upref[tid + 1];
                    add = (tid >= off) ? upref[tid + 1 - off] : 0u;
                }
                __syncthreads();
                if (tid < kSub) upref[tid + 1] = val + add;
                __syncthreads();
            }
            // A wavefront asks the first 64 views about a unit in one round, a view per lane; the views behind them -- 8 of
            // a batch of 72 -- would fill an eighth of a second round, so that round is asked for G = 64 / (nall - 64)
            // units at once (lane = unit slot * (nall - 64) + view - 64), ahead of the units' own rounds: 1 1/8 verdict
            // rounds per unit instead of 2 (the rounds are latency chains, and what this kernel's time is made of).
            const uint32_t per1 = sj.uj.nall > 64 ? (uint32_t)sj.uj.nall - 64u : 0u;
            // (round 5: no more units per round than an even deal over the wavefronts asks for -- a round is a chain of
            // its units' verdicts in ONE wavefront, and with 8 per round the reference's own configuration, 18 480 units
            // on 4 096 wavefronts, kept 2 310 of them busy with eight units each while the others had none; and a bulky
            // object's 56 500 went round twice with eight where twice seven is an even deal)
            const uint32_t gmax = per1 ? 64u / per1 : 1u;  // (nall <= 128: per1 <= 64, gmax >= 1)
            const uint32_t rounds = (total + nworkers * gmax - 1u) / (nworkers * gmax);
            const uint32_t G = min(gmax, max(1u, (total + nworkers * rounds - 1u) / (nworkers * rounds)));
            const uint32_t nbricks_all = sj.lb.nbricks;
            ViewDesc first;  // one descriptor per lane, once per wavefront (it was loaded per unit until round 6: a memory
            first.cmask = nullptr;  // round trip in front of every unit's verdict chain)
            if ((int)lane < sj.uj.nall && (int)lane >= sj.uj.ndense) first = sj.uj.views[lane];
            for (uint32_t i0 = (blockIdx.x * kFlagWaves + wave) * G; i0 < total; i0 += nworkers * G) {
                const uint32_t slot = per1 ? lane / per1 : 0u;
                const uint32_t i = i0 + min(slot, G - 1u);
                const bool has = slot < G && i < total;
                // unit i: one of the bulk list's (the first nb), or unit (i - nb) & 3 of a failed candidate
                uint32_t my_unit = 0u, lo = 0u;
                if (has && i < nb) {
                    uint32_t hi = kSub;  // largest s with upref[s] <= i (per lane: the lanes of a slot agree)
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (upref[mid] <= i) lo = mid; else hi = mid;
                    }
                    my_unit = sj.uj.units[(size_t)lo * sj.uj.cap + (i - upref[lo])];
                } else if (has) {
                    const uint32_t lbq = sj.lb.late[nbricks_all - 1u - ((i - nb) >> 2)];
                    my_unit = lbq * 4u + ((i - nb) & 3u);
                    lo = (lbq * 0x9E3779B1u) >> 24;  // the sub-list its items and survivors go to (as brick_voxels picks it)
                }
                uint32_t v1 = 8u;  // no such view
                if (has && per1) {
                    const int vi = 64 + (int)(lane - slot * per1);
                    const ViewDesc d = sj.uj.views[vi];  // one descriptor per lane
                    const uint32_t lb = my_unit >> 2, w = my_unit & 3u;
                    const uint32_t per_plane = sj.uj.bricks_y * sj.uj.bricks_z;
                    const uint32_t il = lb / per_plane, rem = lb - il * per_plane;
                    const uint32_t by = rem / sj.uj.bricks_z, bz = rem - by * sj.uj.bricks_z;
                    const int j0 = (int)(by * kBrickY), kb = (int)(bz * kBrickZ + w * 16u);
                    const float x = g.ox + (float)(int)(il * g.istride + g.i0) * g.vs;
                    v1 = d.cmask != nullptr ? rect_verdict_cells(d, g, x, j0, j0 + kBrickY - 1, kb, kb + 15) : 0u;
                }
                const unsigned long long e1 = __ballot(v1 == 1u), f1 = __ballot(v1 == 2u), n1 = __ballot(v1 == 0u);
                const unsigned long long word = per1 >= 64u ? ~0ull : ((1ull << per1) - 1ull);
                for (uint32_t u = 0; u < G && i0 + u < total; ++u) {  // wave-uniform
                    const uint32_t src = u * per1;  // the first lane of the slot (lane 0 when there is no second word)
                    const uint32_t unit = __builtin_amdgcn_readlane(my_unit, src);
                    const uint32_t sub = __builtin_amdgcn_readlane(lo, src);
                    const SecondWord second{(e1 >> src) & word & (per1 ? ~0ull : 0ull), (f1 >> src) & word & (per1 ? ~0ull : 0ull),
                                            (n1 >> src) & word & (per1 ? ~0ull : 0ull)};
                    unit_verdicts(sj.uj, g, ctl, unit, sub, lane, second, first);
                }
            }
        }
    }
    if (sj.lb.late != nullptr) {
        // bricks some later view does not keep whole after all: every view, one wavefront per unit
        const uint32_t nlate = ctl->nlate;
        for (uint32_t t = blockIdx.x * kFlagWaves + wave; t < nlate * 4u; t += nworkers) {
            const uint32_t unit = sj.lb.late[t >> 2] * 4u + (t & 3u);
            if (sj.lb.fresh) late_unit<true>(labels, g, sj.lb.allviews, sj.lb.nall, sj.lb.init, unit, sj.lb.bricks_y, sj.lb.bricks_z, lane);
            else late_unit<false>(labels, g, sj.lb.allviews, sj.lb.nall, sj.lb.init, unit, sj.lb.bricks_y, sj.lb.bricks_z, lane);
        }
    }
    if (!overflow) return;
    // The dense fallback.  Every label it reads has been written: the dense stage stores the voxels of the live
    // bricks whatever their state; settled bricks (their fill comes with the list kernels' store blocks, which run
    // whatever the flag says) and late bricks (above, over ALL views, perhaps by a block that has not written
    // them yet) are not this pass's voxels.
    const Append none{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
    const uint64_t nblk = (g.ngroups + kThreads - 1) / kThreads;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const uint64_t grp = blk * kThreads + tid;
        bool skip = grp >= g.ngroups;
        if (!skip && sj.flags != nullptr) {
            Vox4 vx;
            decode_group(g, grp, vx);
            const uint32_t col = (uint32_t)(vx.elem / g.nzp), il = col / g.ny, j = col - il * g.ny;
            const uint32_t fl = sj.flags[(il * sj.bricks_y + j / kBrickY) * sj.bricks_z + vx.k0 / kBrickZ];
            skip = fl != 0u;  // 1 EMPTY, 2 FULL, 4 dead, 5 late, 6 UNTOUCHED (candidates have their answer by now)
        }
        // (a wavefront's lanes leave carve_group's view loop together: skipped lanes still vote)
        if (!skip) {
            const int4 pre = *reinterpret_cast<const int4 *>(labels + grp * 4);
            carve_group<false, true>(labels, g, sj.rest, sj.nrest, 0, grp, pre, none);
        }
    }
}
