// Part of spacecarve.hip (included there, behind sc_engine.h): what a batch of pending views becomes on the stream --
// FusedPlan / flush() (the seven launches of a fused carve, the averaging launches), the stream pool, the device half of an
// engine's set-up and create().

namespace {

// What a fused carve of `nv` views will look like (see flush): decided before anything is launched,
// because a deferred batch is packed according to it.
struct FusedPlan {
    int ndense, nstage1, s1, flag_views;
    bool compact, brick, defer_stores;
    uint32_t bys, bzs, nbricks, nstrips, dense_store_strips;
};

FusedPlan fused_plan(const sc_engine *e, size_t nv, bool has_occ) {
    FusedPlan p{};
    p.ndense = (int)e->dense_views;
    p.nstage1 = (int)e->stage1_views;
    p.compact = e->compact && nv >= (size_t)kMinFusedViews && nv > (size_t)p.ndense &&
                (uint64_t)e->npitch < 0x80000000ull;
    p.bys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY);
    p.bzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
    p.brick = (nv > 1 || e->view_brick) && e->brick && p.bzs <= 64 && (uint64_t)e->npitch < 0x80000000ull &&
              (uint64_t)e->planes * p.bys * p.bzs < 0x40000000ull && has_occ;  // brick ids carry two flag bits in the fill list
    p.nbricks = p.brick ? (uint32_t)((uint64_t)e->planes * p.bys * p.bzs) : 0u;
    p.flag_views = (int)nv;  // every view of the batch may veto a brick, not only the dense stage's
    if (e->flag_views > 0 && e->flag_views < (int64_t)p.flag_views) p.flag_views = (int)e->flag_views;
    p.s1 = (int)std::min<size_t>(nv, (size_t)p.ndense + (size_t)p.nstage1);
    // the -1 fill of empty bricks rides along with the list stages when there are any: strips
    // [0, dense_store_strips) are filled by the dense kernel's store blocks, the others by the list
    // stages' (defer_share sixteenths of them)
    p.nstrips = p.brick ? (uint32_t)((uint64_t)e->planes * p.bys) : 0u;
    p.dense_store_strips = p.nstrips;
    if (p.brick && p.compact && e->defer_stores > 0 && e->defer_share > 0)
        p.dense_store_strips = (uint32_t)((uint64_t)p.nstrips * (uint64_t)(16 - e->defer_share) / 16u);
    p.defer_stores = p.dense_store_strips < p.nstrips;
    return p;
}

// Launch the first `count` pending views (count == 0: all of them).
int flush(sc_engine *e, size_t count = 0) {
    if (e->pending.empty()) return SC_OK;
    {
        int rch = upload_hostbits(e);
        if (rch) return rch;
    }
    size_t nv = count ? std::min(count, e->pending.size()) : e->pending.size();
    // A device batch whose packing was deferred is packed here, in the order its views will be
    // applied: the views the flags kernel, the dense stage and the first survivor stage need go
    // ahead, the others ride beside the dense stage (brick form).  Any other shape of launch packs
    // the whole batch first, in the order given.
    bool ordered = false;
    PackJob ride;
    memset(&ride, 0, sizeof ride);
    uint32_t ride_blocks = 0;
    int packed_ahead = (int)nv;
    if (e->deferred.on) {
        const bool whole = nv == e->pending.size() && nv == (size_t)e->deferred.V && e->mode == SC_MODE_CARVE && nv > 1;
        if (!whole) {
            int rcd = materialize_deferred(e);
            if (rcd) return rcd;
        } else {
            int rcd = step_begin(e);
            if (rcd) return rcd;
            std::vector<uint32_t> perm;
            if (e->view_order == 1) order_views(e->pending, &perm);
            else { perm.resize(nv); for (size_t q = 0; q < nv; ++q) perm[q] = (uint32_t)q; }
            ordered = true;
            PackJob pj;
            rcd = deferred_job(e, &pj);
            if (rcd) return rcd;
            e->deferred.on = false;
            pj.use_order = 1;
            for (size_t q = 0; q < nv; ++q) pj.order[q] = (uint16_t)perm[q];
            const FusedPlan fp = fused_plan(e, nv, true);
            int ahead = (int)nv;
            // (not when the dense kernel fills a share of the strips itself: with riders the flags kernel leaves FULL
            // candidates open until the confirm kernel -- behind the dense kernel -- and only the list stages' store
            // blocks come after that.  Round 4 tried both together for the fill's sake: 4 of 2 600 fuzz cases, all with
            // SC_OPT_DEFER_SHARE 5, kept bricks that nobody filled.)
            if (e->pack_ride && fp.brick && fp.compact && fp.defer_stores && fp.dense_store_strips == 0)
                ahead = std::min<int>((int)nv, std::max(fp.flag_views, fp.s1));
            pj.slot0 = 0;
            pj.nslots = ahead;
            LaunchTimer ltp{e, SC_KERNEL_PACK};
            rcd = ltp.begin();
            if (rcd) return rcd;
            rcd = launch_pack16(e, pj);
            if (rcd) return rcd;
            rcd = ltp.end();
            if (rcd) return rcd;
            if (ahead < (int)nv) {
                ride = pj;
                ride.slot0 = ahead;
                ride.nslots = (int)nv - ahead;
                int64_t rb = pack16_blocks(e, ride);
                if (rb > 0x3fffffffLL) return fail(SC_ERR_INVALID, "mask batch too large");
                ride_blocks = (uint32_t)rb;
                packed_ahead = ahead;
            }
        }
    }
    GridDesc g = grid_desc(e);
    uint64_t blocks = (g.ngroups + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffULL) return fail(SC_ERR_INVALID, "grid too large for one launch");
    // (rows are whole 16-byte groups -- the pitch is a multiple of 64 voxels -- so every kernel takes its
    // vector form, VEC = true; the element-wise forms remain in the templates for a layout without padding)
    dim3 grid((uint32_t)blocks), block(kBlock);
    const ViewDesc *vd = nullptr, *vpin = nullptr;
    if (nv > 1) {
        int rcs = step_begin(e);
        if (rcs) return rcs;
    }
    // a single view in brick form goes through the same kernels as a batch: it needs its descriptor
    // in the device array too
    const bool single_brick = nv == 1 && e->mode == SC_MODE_CARVE &&
                              fused_plan(e, nv, e->pending[0].occ != nullptr).brick;
    if (nv > 1 || single_brick) {
        if (!ordered && e->mode == SC_MODE_CARVE && e->view_order == 1 && nv == e->pending.size())
            order_views(e->pending);
        // descriptor ring: slots are reused only after a wrap, which waits for the stream
        if (nv > e->views_cap || e->views_head + nv > e->views_cap) {
            HIP_TRY(schost::wait_stream(e->stream));
            e->views_head = 0;
        }
        if (nv > e->views_cap) {
            if (e->views_dev) (void)hipFree(e->views_dev);
            if (e->views_pin) (void)hipHostFree(e->views_pin);
            e->views_dev = e->views_pin = nullptr;
            e->views_cap = 0;
            size_t cap = std::max<size_t>(nv * 8, 4096);  // (a wrap every 56 batches of 72 views; 1024 until round 4)
            HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->views_dev), cap * sizeof(ViewDesc)));
            HIP_TRY(sc_pin_malloc(reinterpret_cast<void **>(&e->views_pin), cap * sizeof(ViewDesc),
                                  hipHostMallocDefault));
            e->views_cap = cap;
        }
        ViewDesc *pin = e->views_pin + e->views_head, *dev = e->views_dev + e->views_head;
        memcpy(pin, e->pending.data(), nv * sizeof(ViewDesc));
        e->views_head += nv;
        vd = dev;
        vpin = pin;
    }
    const ViewDesc &one = e->pending[0];
    int rc;
    // the descriptors reach the device array either by a copy on the stream, or -- brick form of
    // the fused carve -- through the flags kernel, which gets its own in its arguments
    bool desc_uploaded = false;
    auto upload_desc = [&]() -> int {
        if (desc_uploaded || vd == nullptr) return SC_OK;
        desc_uploaded = true;
        HIP_TRY(hipMemcpyAsync(const_cast<ViewDesc *>(vd), vpin, nv * sizeof(ViewDesc), hipMemcpyHostToDevice, e->stream));
        return SC_OK;
    };
    if (e->mode == SC_MODE_CARVE) {
        int32_t *st = static_cast<int32_t *>(e->state);
        int32_t init = init_bits_i32(e);
        // fused carve with survivor compaction: dense for the first `ndense` views, then lists
        const FusedPlan fp = fused_plan(e, nv, one.occ != nullptr);
        const int ndense = fp.ndense, nstage1 = fp.nstage1, flag_views = fp.flag_views;
        const uint32_t list_blocks = (uint32_t)e->list_blocks;
        const bool compact = fp.compact, brick = fp.brick, defer_stores = fp.defer_stores;
        // what sc_values_sparse may take from this launch's verdict bytes and lists (sc_sparse.h)
        e->sparse_exact = brick && e->fresh;
        e->sparse_late = false;
        Append ap{nullptr, nullptr, 0u, 0u, nullptr, 0u, 0u};
        int dense_views = (int)nv;
        const uint32_t bys = fp.bys, bzs = fp.bzs, nbricks = fp.nbricks, nstrips = fp.nstrips;
        const uint32_t dense_store_strips = fp.dense_store_strips;
        // strips set to -1 ahead of the verdicts, by fill blocks in front of the flags kernel's own (SpecFill): a fresh
        // volume whose fill is all the list stages' (so that everything behind the flags kernel that writes labels
        // comes later on the stream)
        uint32_t spec_strips = 0;
        if (fp.brick && fp.compact && fp.defer_stores && dense_store_strips == 0 && e->fresh && e->spec_share > 0)
            spec_strips = (uint32_t)((uint64_t)fp.nstrips * (uint64_t)e->spec_share / 16u);
        const bool desc_by_flags = brick && flag_views <= kFlagWaves;
        if (!desc_by_flags) {
            rc = upload_desc();
            if (rc) return rc;
        }
        if (compact || brick) {
            rc = ensure_ctl(e);
            if (rc) return rc;
            // list counters, overflow flag, live-brick count: this batch takes the block the previous
            // batch's final stage left zeroed (a memset only if there was no such stage)
            if (compact) {
                e->ctl_idx ^= 1;
                e->ctl = e->ctl2[e->ctl_idx];
                if (!e->ctl_clean[e->ctl_idx]) HIP_TRY(hipMemsetAsync(e->ctl, 0, sizeof(ListCtl), e->stream));
            }
            // (a launch without survivor stages keeps the block: its two counters alternate, see ListCtl)
            e->ctl_clean[e->ctl_idx] = false;
        }
        const uint32_t parity = (uint32_t)(e->flag_launches & 1u);
        // blocks of the flags kernel per sub-list of the candidate list (ListCtl::ncand): sub-list s holds the candidates of
        // blocks [s per, (s + 1) per) at cands + s per 64 -- at most the bricks of those blocks, so the lists fit in nbricks words
        const uint32_t cand_per = std::max<uint32_t>(1u, (uint32_t)(((nbricks + 63u) / 64u + kCandSub - 1) / kCandSub));
        if (brick) ++e->flag_launches;
        if (compact) {
            rc = ensure_lists(e);
            if (rc) return rc;
            ap.list = e->lists;
            ap.ctl = e->ctl;
            ap.subcap = e->subcap;
            dense_views = ndense;
        }
        // bulk units: brick form with survivor stages, every view with its cell level
        bool bulk_on = compact && brick && e->bulk_min > 0 && e->bulk != nullptr && e->items != nullptr;
        if (nv > 128) bulk_on = false;  // the units' verdict masks cover 128 views
        for (size_t q = 0; q < nv && bulk_on; ++q) bulk_on = e->pending[q].cmask != nullptr;
        if (bulk_on) {
            ap.bulk = e->bulk;
            ap.bulkcap = e->bulkcap;
            ap.bulk_min = (uint32_t)e->bulk_min;
        }
        e->last_bulk = bulk_on;
        LaunchTimer lt{e, SC_KERNEL_CARVE};
        if (!brick) {  // the brick form starts the timer after its flags kernel
            rc = lt.begin();
            if (rc) return rc;
        }
        if (nv == 1 && !brick) {
            // kStreamGroups groups per lane when the state is streamed through (see kernel)
            uint32_t per_block = !e->fresh ? kBlock * kStreamGroups : kBlock;
            dim3 grid1((uint32_t)((g.ngroups + per_block - 1) / per_block));
#define LAUNCH_CARVE1(F, V) \
    hipLaunchKernelGGL((carve_kernel_1<F, V>), grid1, block, 0, e->stream, st, g, one, init)
            if (e->fresh) {
                LAUNCH_CARVE1(true, true);
            } else {
                LAUNCH_CARVE1(false, true);
            }
#undef LAUNCH_CARVE1
        } else {
            if (brick) {
                // live-list walkers (whole groups of 8 XCDs), then store blocks, then packing riders; with
                // riders the walkers leave wavefront slots free for them
                const uint32_t nwalkers = ((uint32_t)(ride_blocks ? e->brick_walkers : e->list_blocks) + 7u) & ~7u;
                dim3 bgrid(nwalkers + dense_store_strips + ride_blocks);
                if (!e->dead) {
                    HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->dead), (size_t)nbricks));
                    e->dead_clean = false;
                }
                const int dead_stale = e->dead_clean ? 0 : 1;  // the flags kernel rewrites them all
                e->dead_clean = true;
                // unit verdicts (cell level) by the views packed ahead, inside the dense stage
                int nverd = 0;
                const uint32_t verd_max_live = e->unit_cull == 2 ? 0xffffffffu : (uint32_t)(nbricks / 2);
                const uint32_t bulk_min_live = (uint32_t)((uint64_t)nbricks * (uint64_t)e->bulk_live / 16u);
                if (compact && e->unit_cull) {
                    nverd = std::min(packed_ahead, 16);
                    for (int q = 0; q < nverd; ++q)
                        if (e->pending[(size_t)q].cmask == nullptr) nverd = 0;
                }
                LaunchTimer ltf{e, SC_KERNEL_FLAGS};
                rc = ltf.begin();
                if (rc) return rc;
                FlagViews own{};
                DescCopy dc{nullptr, nullptr, 0u};
                if (desc_by_flags) {
                    for (int q = 0; q < flag_views; ++q) own.v[q] = e->pending[(size_t)q];
                    dc = DescCopy{reinterpret_cast<const uint32_t *>(vpin),
                                  reinterpret_cast<uint32_t *>(const_cast<ViewDesc *>(vd)),
                                  (uint32_t)(nv * sizeof(ViewDesc) / 4)};
                    desc_uploaded = true;
                }
                SpecFill sf{nullptr, 0u, 0u};
                if (spec_strips > 0) {
                    // strip s starts at column (s / bys) * ny + (s % bys) * 16; the columns are contiguous rows of nzp labels
                    const uint64_t cols = (uint64_t)(spec_strips / bys) * (uint64_t)e->ny + (uint64_t)(spec_strips % bys) * kBrickY;
                    sf = SpecFill{st, cols * (uint64_t)e->nzp * 4u, (uint32_t)e->spec_blocks};
                }
                hipLaunchKernelGGL(brick_flags_kernel, dim3(sf.nblocks + (nbricks + 63u) / 64u), dim3(64 * kFlagWaves), 0,
                                   e->stream, g, desc_by_flags ? static_cast<const ViewDesc *>(nullptr) : vd,
                                   flag_views, bys, bzs, nbricks, e->flags, e->live, e->ctl, own, dc,
                                   desc_by_flags ? vpin : vd, e->full_bricks ? packed_ahead : 0, (int)nv, e->dead,
                                   dead_stale, parity, compact ? static_cast<uint32_t *>(nullptr) : e->fill_list, sf,
                                   compact ? e->fill_list : static_cast<uint32_t *>(nullptr), cand_per);  // (the room of the fill list holds the candidate list when nothing fills from a list)
                e->last_parity = parity;
                rc = ltf.end();
                if (rc) return rc;
                rc = lt.begin();  // SC_KERNEL_CARVE times the dense kernel alone
                if (rc) return rc;
                if (!compact) {
                    // no survivor stages: walkers on the live list, fillers on the fill list
                    const dim3 lgrid(nwalkers + (uint32_t)std::max<int64_t>(e->fill_blocks, 64));
                    if (e->fresh)
                        hipLaunchKernelGGL((carve_brick_light_kernel<true>), lgrid, block, 0, e->stream, st, g, vd,
                                           dense_views, init, bys, bzs, e->live, e->fill_list, e->ctl, nwalkers, parity);
                    else
                        hipLaunchKernelGGL((carve_brick_light_kernel<false>), lgrid, block, 0, e->stream, st, g, vd,
                                           dense_views, init, bys, bzs, e->live, e->fill_list, e->ctl, nwalkers, parity);
                } else {
                    // (every dense view certified by the host: the instance without the general projection path)
                    // (a thinned-out unit may take one more pair of the views packed ahead: brick_voxels)
                    const int nextra = e->dense_extra ? std::max(0, std::min(2, packed_ahead - dense_views)) : 0;
                    bool dense_safe = e->safe_kernels != 0;
                    for (int q = 0; q < dense_views + nextra && dense_safe; ++q) dense_safe = e->pending[(size_t)q].safe != 0;
#define LAUNCH_BRICK(F, S)                                                                                          \
    hipLaunchKernelGGL((carve_brick_kernel<F, S>), bgrid, block, 0, e->stream, st, g, vd, dense_views, init, ap, bys, bzs, \
                       e->flags, e->live, e->ctl, nwalkers, dense_store_strips, ride, pack_form(e, ride), parity, nverd,    \
                       verd_max_live, bulk_min_live, nextra)
                    if (e->fresh && dense_safe) LAUNCH_BRICK(true, true);
                    else if (e->fresh) LAUNCH_BRICK(true, false);
                    else if (dense_safe) LAUNCH_BRICK(false, true);
                    else LAUNCH_BRICK(false, false);
#undef LAUNCH_BRICK
                }
            } else {
#define LAUNCH_CARVE(F, V)                                                                    \
    hipLaunchKernelGGL((carve_kernel<F, V>), grid, block, 0, e->stream, st, g, vd, dense_views, \
                       init, ap)
                if (e->fresh) {
                    LAUNCH_CARVE(true, true);
                } else {
                    LAUNCH_CARVE(false, true);
                }
#undef LAUNCH_CARVE
            }
        }
        HIP_TRY(hipGetLastError());
        rc = lt.end();
        if (rc) return rc;
        if (compact) {
            int s1 = (int)std::min<size_t>(nv, (size_t)ndense + nstage1);
            uint32_t *l0 = e->lists, *l1 = e->lists + (size_t)kSub * e->subcap;
            LaunchTimer lt2{e, SC_KERNEL_LIST};
            rc = lt2.begin();
            if (rc) return rc;
            int vg = (int)e->view_group;
            // open FULL candidates exist only when packing rode beside the dense stage
            CullStores none{nullptr, 0u, 0u, 0u, 0u, 0, 0, 0u, 0, 0u}, cs = none;
            if (ride_blocks) {
                e->sparse_late = true;
                // the riders have packed the rest of the masks: open FULL candidates get their answer
                // (a block per 64 entries of the candidate list, a persistent grid of at most 4096; without candidates
                // every block leaves after eight scalar loads)
                const uint32_t nconfirm = std::min<uint32_t>((nbricks + 63u) / 64u, 4096u);
                // (a candidate that fails takes the bulk units' road when the batch has a bulk list: UnitRoad)
                const UnitRoad road{(bulk_on && e->late_road) ? st : nullptr, init, e->fresh ? 1 : 0, nbricks};
                hipLaunchKernelGGL(brick_confirm_kernel, dim3(nconfirm), dim3(64 * kConfirmWaves), 0, e->stream, g, vd,
                                   packed_ahead, (int)nv, bys, bzs, e->flags, e->fill_list, cand_per, e->late, e->ctl, parity, road);
            }
            // Too few bulk units for their verdicts are taken by the first survivor stage as they are (UnitSpill); a
            // batch with a single (final) list stage has no such stage: its units are always asked
            const uint32_t unit_floor = (size_t)ndense + (size_t)nstage1 >= nv ? 0u : (uint32_t)e->bulk_floor;
            {
                // bulk units, late bricks, the dense fallback, the next batch's counters: one launch, always there
                // (what it finds to do is decided on the device)
                SpecialJob sj;
                memset(&sj, 0, sizeof sj);
                if (bulk_on)
                    sj.uj = UnitJob{e->bulk, e->bulkcap, e->items, e->itemcap, vd, (int32_t)nv, ndense, bys, bzs, st,
                                    e->lists, e->subcap, (uint32_t)e->item_bias, unit_floor};
                sj.lb = LateBricks{ride_blocks ? e->late : nullptr, nbricks, vd, e->flags, (int32_t)nv, init, e->fresh ? 1 : 0, bys, bzs};
                sj.next = e->ctl2[e->ctl_idx ^ 1];
                sj.rest = vd + ndense;
                sj.nrest = (int32_t)nv - ndense;
                sj.flags = brick ? e->flags : nullptr;
                sj.bricks_y = bys;
                sj.bricks_z = bzs;
                hipLaunchKernelGGL(carve_special_kernel, dim3((uint32_t)e->unit_blocks), dim3(64 * kFlagWaves), 0,
                                   e->stream, st, g, e->ctl, sj);
                e->ctl_clean[e->ctl_idx ^ 1] = true;
            }
            // final stage with deferred stores: e->defer_stores persistent list blocks (they leave
            // wavefront slots free) and one short store block per strip behind them
            dim3 fgrid(list_blocks);
            CullStores cs1 = none;
            dim3 grid1(list_blocks);
            if (defer_stores) {
                // the first list stage may take a share of the fill as well (it waits on memory)
                const uint32_t first = std::max(dense_store_strips, spec_strips);  // (one of the two is 0)
                uint32_t mid = first;
                if ((size_t)s1 < nv && e->stage1_store_share > 0) {
                    mid += (uint32_t)((uint64_t)(nstrips - first) * (uint64_t)e->stage1_store_share / 16u);
                    const uint32_t n1 = mid - first, f1 = std::min<uint32_t>((uint32_t)e->fill_blocks, n1);
                    cs1 = CullStores{e->flags, bys, bzs, mid, first, init == 0 ? 1 : init, e->fresh ? 1 : 0, f1, init, 0u};
                    grid1 = dim3((uint32_t)e->stage1_list_blocks + (f1 ? f1 : n1));
                }
                // (the final stage also walks the strips filled ahead, for their FULL / UNTOUCHED bricks)
                const uint32_t nf = nstrips - mid + spec_strips, ff = std::min<uint32_t>((uint32_t)e->fill_blocks, nf);
                cs = CullStores{e->flags, bys, bzs, nstrips, mid, init == 0 ? 1 : init, e->fresh ? 1 : 0, ff, init, spec_strips};
                fgrid = dim3((uint32_t)e->defer_stores + (ff ? ff : nf));
            }
            // every view of the batch certified by the host (certify_view: any real rig): the instances without the general path
            bool all_safe = e->safe_kernels != 0;
            for (size_t q = 0; q < nv && all_safe; ++q) all_safe = e->pending[q].safe != 0;
#define LAUNCH_LIST(FIN, GRID, ...)                                                                      \
    do {                                                                                                 \
        if (all_safe) hipLaunchKernelGGL((carve_list_kernel<FIN, 2, true>), GRID, block, 0, e->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((carve_list_kernel<FIN, 2>), GRID, block, 0, e->stream, __VA_ARGS__);     \
    } while (0)
            // (two survivors per lane in both stages: the instances with one and with four were retired with their
            // knobs in round 6, and so was the optional second stage)
            // stage 1 (l0 -> l1), final stage on what is left
            uint32_t *nolist = nullptr;
            // the final stage also takes the work items of the bulk units
            const UnitItems noitems{nullptr, 0u, nullptr, 0u, 0u}, ui{bulk_on ? e->items : nullptr, e->itemcap, vd, bys, bzs};
            // ... the first one the bulk units of a batch that has too few for their verdicts (decided on the device)
            const UnitSpill nospill{nullptr, 0u, 0u, 0u, 0u}, us{bulk_on ? e->bulk : nullptr, e->bulkcap, unit_floor, bys, bzs};
            if ((size_t)s1 >= nv) {
                LAUNCH_LIST(true, fgrid, st, g, vd + ndense, s1 - ndense, l0, nolist, e->ctl, 0, 0, e->subcap, vg, cs, ui, nospill, s1 - ndense);
            } else {
                LAUNCH_LIST(false, grid1, st, g, vd + ndense, s1 - ndense, l0, l1, e->ctl, 0, 1, e->subcap, vg, cs1, noitems, us, (int)nv - ndense);
                LAUNCH_LIST(true, fgrid, st, g, vd + s1, (int)nv - s1, l1, nolist, e->ctl, 1, 1, e->subcap, vg, cs, ui, nospill, (int)nv - s1);
            }
#undef LAUNCH_LIST
            HIP_TRY(hipGetLastError());
            rc = lt2.end();
            if (rc) return rc;
        }
    } else {
        float *st = static_cast<float *>(e->state);
        rc = upload_desc();
        if (rc) return rc;
        // brick form: uint8 masks with uniformity flags on every view of the batch, a table, a grid it fits
        const uint32_t abys = (uint32_t)((e->ny + kBrickY - 1) / kBrickY), abzs = (uint32_t)((e->nz + kBrickZ - 1) / kBrickZ);
        bool abrick = nv > 1 && e->avg_brick && (uint64_t)e->npitch < 0x80000000ull &&
                      (uint64_t)e->planes * abys * abzs < 0x80000000ull;
        bool any_f32 = false;
        for (size_t q = 0; q < nv && abrick; ++q) {
            const ViewDesc &pd = e->pending[q];
            abrick = (pd.pad == 1 && e->lut_dev != nullptr && pd.occ != nullptr) || (pd.pad == 2 && pd.occ != nullptr);
            any_f32 |= pd.pad == 2;
        }
        if (abrick) {
            const uint32_t anb = (uint32_t)((uint64_t)e->planes * abys * abzs);
            const size_t need = (size_t)anb * nv;
            if (need > e->verd_cap) {
                HIP_TRY(schost::wait_stream(e->stream));
                if (e->verd) (void)hipFree(e->verd);
                e->verd = nullptr;
                e->verd_cap = 0;
                HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->verd), need));
                e->verd_cap = need;
            }
            if (any_f32 && need > e->verdf_cap) {  // the flat values of float32 views
                HIP_TRY(schost::wait_stream(e->stream));
                if (e->verdf) (void)hipFree(e->verdf);
                e->verdf = nullptr;
                e->verdf_cap = 0;
                HIP_TRY(sc_dev_malloc(reinterpret_cast<void **>(&e->verdf), need * 4));
                e->verdf_cap = need;
            }
            uint32_t *verdf = any_f32 ? e->verdf : nullptr;
            LaunchTimer ltf{e, SC_KERNEL_FLAGS};
            rc = ltf.begin();
            if (rc) return rc;
            hipLaunchKernelGGL(avg_flags_kernel, dim3((anb + kBlock - 1) / kBlock, (uint32_t)nv), block, 0, e->stream,
                               g, vd, (int)nv, abys, abzs, anb, e->verd, verdf);
            rc = ltf.end();
            if (rc) return rc;
            LaunchTimer lta{e, SC_KERNEL_AVERAGE};
            rc = lta.begin();
            if (rc) return rc;
            if (e->fresh)
                hipLaunchKernelGGL(average_brick_kernel<true>, dim3(anb), block, 0, e->stream, st, g, vd, (int)nv,
                                   e->default_value, e->lut_dev, abys, abzs, e->verd, verdf);
            else
                hipLaunchKernelGGL(average_brick_kernel<false>, dim3(anb), block, 0, e->stream, st, g, vd, (int)nv,
                                   e->default_value, e->lut_dev, abys, abzs, e->verd, verdf);
            HIP_TRY(hipGetLastError());
            rc = lta.end();
            if (rc) return rc;
            rc = step_end(e, nv > 1);
            if (rc) return rc;
            e->fresh = false;
            e->pending.erase(e->pending.begin(), e->pending.begin() + (ptrdiff_t)nv);
            if (e->pending.empty()) arena_reset(e);
            return SC_OK;
        }
        LaunchTimer lt{e, SC_KERNEL_AVERAGE};
        rc = lt.begin();
        if (rc) return rc;
#define LAUNCH_AVG(F, V)                                                                         \
    do {                                                                                         \
        if (nv == 1)                                                                             \
            hipLaunchKernelGGL((average_kernel_1<F, V>), grid, block, 0, e->stream, st, g, one,  \
                               e->default_value, e->lut_dev);                                    \
        else                                                                                     \
            hipLaunchKernelGGL((average_kernel<F, V>), grid, block, 0, e->stream, st, g, vd,     \
                               (int)nv, e->default_value, e->lut_dev);                           \
    } while (0)
        if (e->fresh) {
            LAUNCH_AVG(true, true);
        } else {
            LAUNCH_AVG(false, true);
        }
#undef LAUNCH_AVG
        HIP_TRY(hipGetLastError());
        rc = lt.end();
        if (rc) return rc;
    }
    rc = step_end(e, nv > 1);
    if (rc) return rc;
    e->fresh = false;
    e->pending.erase(e->pending.begin(), e->pending.begin() + (ptrdiff_t)nv);
    if (e->pending.empty()) arena_reset(e);  // masks of launched views are dead in stream order
    return SC_OK;
}

int after_enqueue(sc_engine *e) {
    if (e->views_per_launch > 0) {
        while ((int64_t)e->pending.size() >= e->views_per_launch) {
            int rc = flush(e, (size_t)e->views_per_launch);
            if (rc) return rc;
        }
        return SC_OK;
    }
    if ((int64_t)e->pending.size() >= e->max_pending) return flush(e);
    return SC_OK;
}

int check_view_args(const sc_engine *e, const float *K, const float *R, const float *t,
                    const void *mask, int H, int W) {
    if (!e) return fail(SC_ERR_INVALID, "null engine");
    if (!K || !R || !t || !mask) return fail(SC_ERR_INVALID, "null view argument");
    // (a view's bits: below 2^32 bytes; the words of a strip of its bit tiles, H rounded up to 32: below 2^24)
    if (H <= 0 || W <= 0 || H > (1 << 24) - 32 || W > (1 << 24) || (int64_t)H * W > ((int64_t)1 << 34))
        return fail(SC_ERR_INVALID, "bad mask shape %d x %d", H, W);
    return SC_OK;
}

// Engine streams are kept between engines (round 5).  Creating a non-blocking stream is a hardware queue's worth of
// set-up in the runtime: 84-139 ms for the first one of a process and, now and then, 6-40 ms for a later one -- the
// "37 ms first batch" of a fresh engine that round 4's bench line showed on the driver's box (SC_TRACE_ALLOC=1 names the
// call).  A destroyed engine's stream (idle: sc_destroy has waited for it) goes on a short per-device list and the next
// engine on that device takes it from there; a process's first engine still pays the first creation, once.
std::mutex g_stream_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;  // (device, idle stream)
constexpr size_t kStreamPoolMax = 16;

// sc_prewarm: a thread that is bringing the runtime up and making the device's first stream; whoever wants a stream of
// that device waits for it (one creation, not two side by side) and finds the stream on the list
std::condition_variable g_prewarm_cv;
int g_prewarm_running[64] = {0};

hipError_t take_stream(int device, hipStream_t *out) {
    {
        std::unique_lock<std::mutex> lk(g_stream_mu);
        if (device >= 0 && device < 64) g_prewarm_cv.wait(lk, [&] { return g_prewarm_running[device] == 0; });
        for (size_t i = 0; i < g_stream_pool.size(); ++i)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (ptrdiff_t)i);
                return hipSuccess;
            }
    }
    return sctrace::timed("hipStreamCreate", __LINE__, 0, [&] { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); });
}

void give_stream_back(int device, hipStream_t s) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (g_stream_pool.size() < kStreamPoolMax) {
            g_stream_pool.emplace_back(device, s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

// The device half of an engine's set-up: the device is there and is a gfx950, a stream, the state.
int device_setup_body(sc_engine *e);
int device_setup(sc_engine *e) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = device_setup_body(e);
    e->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
int device_setup_body(sc_engine *e) {
    const int device = e->device;
    // `device` is a HIP ordinal; only that device has to be a gfx950
    int ndev = 0;
    hipError_t hq = hipGetDeviceCount(&ndev);
    if (hq != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(hq));
    if (device < 0 || device >= ndev)
        return fail(SC_ERR_DEVICE, "device %d not available (%d HIP device(s) visible)", device, ndev);
    {
        hipDeviceProp_t prop;
        hq = sctrace::timed("hipGetDeviceProperties", __LINE__, 0, [&] { return hipGetDeviceProperties(&prop, device); });
        if (hq != hipSuccess) return fail(SC_ERR_DEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(hq));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(SC_ERR_DEVICE, "device %d is %s; this engine is built for gfx950 only", device,
                        prop.gcnArchName);
    }
    hipError_t he = sctrace::timed("hipSetDevice", __LINE__, 0, [&] { return hipSetDevice(device); });
    if (he == hipSuccess) he = take_stream(device, &e->own_stream);
    if (he == hipSuccess) he = sc_dev_malloc(&e->state, (size_t)e->npitch * 4);
    if (he != hipSuccess)
        return fail(he == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_DEVICE, "engine setup failed: %s", hipGetErrorString(he));
    e->stream = e->own_stream;
    return SC_OK;
}

// The engine owns the x-planes  i0, i0 + istride, ...  (`planes` of them) of the global grid.
int create(sc_engine **out, int64_t nx, int64_t ny, int64_t nz, int64_t i0, int64_t istride,
           int64_t planes, const float *origin, float vs, int mode, float default_value, int device, bool deferred = false) {
    if (!out) return fail(SC_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!origin) return fail(SC_ERR_INVALID, "null origin");
    if (nx <= 0 || ny <= 0 || nz <= 0) return fail(SC_ERR_INVALID, "shape must be positive");
    // int -> float of an index must be exact
    if (nx > (1 << 24) || ny > (1 << 24) || nz > (1 << 24))
        return fail(SC_ERR_INVALID, "axis longer than 2^24 voxels");
    if (i0 < 0 || istride < 1 || planes < 1 || i0 + (planes - 1) * istride >= nx)
        return fail(SC_ERR_INVALID, "bad slab / plane set (first %lld, stride %lld, planes %lld of %lld)",
                    (long long)i0, (long long)istride, (long long)planes, (long long)nx);
    if (mode != SC_MODE_CARVE && mode != SC_MODE_AVERAGE)
        return fail(SC_ERR_INVALID, "unknown mode %d", mode);
    sc_engine *e = new (std::nothrow) sc_engine();
    if (!e) return fail(SC_ERR_NOMEM, "host allocation failed");
    e->device = device;
    e->mode = mode;
    e->nx = nx; e->ny = ny; e->nz = nz; e->i0 = i0; e->istride = istride; e->planes = planes;
    e->n = planes * ny * nz;
    e->nzp = (nz + 63) / 64 * 64;
    e->npitch = planes * ny * e->nzp;
    memcpy(e->origin, origin, sizeof e->origin);
    e->vs = vs;
    e->default_value = default_value;
    e->fresh = true;
    if (deferred) {
        // the device half on a thread of its own: the caller goes on (reads its files, decodes them) and the first
        // call that needs the device joins -- and takes the failure, if there is one
        e->setup_pending = true;
        try {
            e->setup_thread = std::thread([e]() {
                e->setup_rc = device_setup(e);
                if (e->setup_rc != SC_OK) e->setup_err = g_err;  // (the thread's own message)
            });
        } catch (...) {
            e->setup_pending = false;
            int rc = device_setup(e);
            if (rc) {
                sc_destroy(e);
                return rc;
            }
        }
        *out = e;
        return SC_OK;
    }
    int rc = device_setup(e);
    if (rc) {
        sc_destroy(e);
        return rc;
    }
    *out = e;
    return SC_OK;
}

}  // namespace
