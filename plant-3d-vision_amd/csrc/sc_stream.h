// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- where survivors go (Append), carve_group and the dense kernel without bricks (backprojection.c:57-84).

// Where a fused launch appends the voxels that are still alive after its dense views.
struct Append {
    uint32_t *list;   // nullptr: no append
    ListCtl *ctl;
    uint32_t subcap;  // entries per sub-list
    uint32_t sub;     // sub-list of this block
    // brick form: a wavefront's share of a brick (16 columns x 16 voxels, a UNIT) with at least `bulk_min`
    // voxels alive after the dense views goes on the bulk list as a whole instead of voxel by voxel
    // (unit_verdicts: the views are asked about the unit, one view per lane, before any projects its voxels)
    uint32_t *bulk;   // nullptr: no such list
    uint32_t bulkcap; // units per sub-list
    uint32_t bulk_min;
};

// carve (backprojection.c:57-84) of one 4-voxel group over views[0..nviews), state in
// registers.  FRESH: the state is known to be `init` everywhere (nothing applied since
// create/clear) and is not read.  VEC: nz % 4 == 0, state accessed as int4 (`pre` holds the
// group's state, already loaded by the caller so that loads of several groups overlap).
template <bool FRESH, bool VEC>
__device__ __forceinline__ void carve_group(int32_t *__restrict__ labels, const GridDesc &g,
                                            const ViewDesc *__restrict__ views, int nviews,
                                            int32_t init, uint64_t grp, int4 pre,
                                            const Append &ap) {
    Vox4 vx;
    int32_t lab[4], was[4];
    // a grid whose rows are padded (nz not a multiple of 64) has groups that own fewer than four voxels:
    // they are told apart up front; on an unpadded grid a group is decoded only if something in it lives
    const bool padded = g.nzp != g.nz;  // grid-uniform
    if (!VEC || padded) decode_group(g, grp, vx);
    int32_t *p = labels + (VEC ? grp * 4 : vx.elem);
    if (FRESH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) lab[e] = init;
    } else if (VEC) {
        lab[0] = pre.x; lab[1] = pre.y; lab[2] = pre.z; lab[3] = pre.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) lab[e] = (e < (int)vx.nvalid) ? p[e] : -1;
    }
    uint32_t alive = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (VEC && padded && e >= (int)vx.nvalid) lab[e] = -1;  // padding counts as carved
        was[e] = lab[e];
        if ((VEC || e < (int)vx.nvalid) && lab[e] != -1) alive |= 1u << e;  // :67
    }
    if (!FRESH && alive == 0) return;  // nothing to do and nothing to write
    if (VEC && !padded) decode_group(g, grp, vx);

    float z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = g.oz + (float)(int)(vx.k0 + e) * g.vs;  // :73

    for (int vi = 0; vi < nviews; ++vi) {
        if (__ballot(alive != 0) == 0) break;  // whole wavefront carved
        const ViewDesc d = views[vi];          // wave-uniform: scalar loads, once per view
        float ax = d.R[0] * vx.x + d.R[1] * vx.y;
        float ay = d.R[3] * vx.x + d.R[4] * vx.y;
        float az = d.R[6] * vx.x + d.R[7] * vx.y;
        const uint32_t *bits = static_cast<const uint32_t *>(d.mask);
        bool ok[4];
        uint32_t w[4];
        int sh[4];
        // all four projections first, then the four gathers in flight together
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int u, v;
            ok[e] = project(ax, ay, az, z[e], d, u, v) & ((alive >> e) & 1u);
            sh[e] = u & 31;
            // unconditional gather (word 0 when the voxel is out): no branch per element
            w[e] = load_mask_word(bits, ok[e] ? mask_word_index(u, v, d.strip) : 0u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ok[e]) {
                if (((w[e] >> sh[e]) & 1u) == 0) {  // :79
                    lab[e] = -1;
                    alive &= ~(1u << e);
                } else if (lab[e] == 0) {  // :81
                    lab[e] = 1;
                }
            }
        }
    }

    if (VEC) {
        bool changed = FRESH || lab[0] != was[0] || lab[1] != was[1] || lab[2] != was[2] ||
                       lab[3] != was[3];
        if (changed) *reinterpret_cast<int4 *>(p) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < (int)vx.nvalid && (FRESH || lab[e] != was[e])) p[e] = lab[e];
    }

    if (ap.list != nullptr) {
        // survivors -> sub-list `ap.sub`: one atomic per wavefront, entries = slab-local voxel
        // index, bit 31 = "label is still 0" (a later foreground hit must write 1)
        unsigned long long b[4];
        uint32_t total = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b[e] = __ballot((alive >> e) & 1u);
            total += (uint32_t)__popcll(b[e]);
        }
        if (total != 0) {  // wave-uniform
            unsigned long long act = __ballot(1);
            uint32_t lane = __lane_id();
            uint32_t base = 0;
            if (lane == (uint32_t)(__ffsll((long long)act) - 1))
                base = atomicAdd(&ap.ctl->count[0][ap.sub].n, total);
            base = __shfl(base, __ffsll((long long)act) - 1);
            if (base + total > ap.subcap) {
                if (lane == (uint32_t)(__ffsll((long long)act) - 1)) ap.ctl->overflow = 1u;
            } else {
                uint32_t *dst = ap.list + (size_t)ap.sub * ap.subcap + base;
                unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                uint32_t off = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((alive >> e) & 1u) {
                        uint32_t rank = off + (uint32_t)__popcll(b[e] & below);
                        dst[rank] = (uint32_t)(vx.elem + e) | (lab[e] == 0 ? 0x80000000u : 0u);
                    }
                    off += (uint32_t)__popcll(b[e]);
                }
            }
        }
    }
}

// A chunk of views per launch: one group per lane; optionally appends the survivors.
template <bool FRESH, bool VEC>
__global__ __launch_bounds__(kBlock) void carve_kernel(int32_t *__restrict__ labels, GridDesc g,
                                                       const ViewDesc *__restrict__ views,
                                                       int nviews, int32_t init, Append ap) {
    uint32_t lb = spread_block(blockIdx.x, gridDim.x);
    uint64_t grp = (uint64_t)lb * kBlock + threadIdx.x;
    if (grp >= g.ngroups) return;
    int4 pre = make_int4(0, 0, 0, 0);
    if (!FRESH && VEC) pre = *reinterpret_cast<const int4 *>(labels + grp * 4);
    ap.sub = (lb * 0x9E3779B1u) >> 24;  // kSub == 256: hashed, so a dense region loads every sub-list alike
    carve_group<FRESH, VEC>(labels, g, views, nviews, init, grp, pre, ap);
}
