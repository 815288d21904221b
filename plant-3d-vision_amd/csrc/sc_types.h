// Part of spacecarve.hip (included there, inside its anonymous namespace, in this order: sc_types, sc_project,
// sc_stream, sc_pack, sc_verdicts, sc_bricks, sc_lists, sc_average, sc_misc) -- the layouts both sides share: view and grid descriptors, the control block of the survivor lists.


struct ViewDesc {   // 128 bytes, read with scalar loads (the view index is wave-uniform)
    float K[4];     // fx fy cx cy
    float R[9];     // row-major
    float t[3];
    const void *mask;  // carve: bit words, 32 x 32-pixel tiles strip by strip (tile (tx, ty) at word tx * strip + 32 ty); average: tiled bytes / floats
    int32_t W, H;
    int32_t tiles_x;
    int32_t pad;
    float Wf, Hf;
    const uint8_t *occ;  // carve: one byte per 32x32 tile: bit 0 some foreground, bit 1 only foreground
    int32_t safe;        // certify_view(): every voxel centre of the grid has 2^-10 < pz and |px|, |py|, pz < 2^30
                         // under this pose, and the intrinsics are finite and below 2^30 (see project())
    int32_t strip;       // carve: the words of a 32-pixel-wide STRIP of the bit tiles = 32 x tile rows (mask_byte_offset)
    const uint32_t *cmask;  // carve, 16-byte pack form: per 32x32 tile the 4x4 map of its 8x8-pixel CELLS -- bits 0..15
                            // "cell holds some foreground", bits 16..31 "cell holds some background" (cell (cx, cy) of
                            // the tile at bit cy * 4 + cx; padding counts as background); null: no cell level
    uint64_t reserved;
};
static_assert(sizeof(ViewDesc) == 128, "ViewDesc layout");
// Descriptor `i` (wave-uniform) by SCALAR loads, whatever the compiler knows about the pointer: through the constant
// address space.  A pointer that reaches a kernel inside a struct carries no no-alias promise, so its loads are kept
// coherent with the kernel's own stores -- seven vector loads per lane for a descriptor (round 4: 9 M of them per
// dispatch in the final stage of the dense scene, against 2.4 M mask gathers).  No voxel kernel writes a descriptor.
__device__ __forceinline__ ViewDesc scalar_desc(const ViewDesc *views, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) ViewDesc *const_views_t;
    return ((const_views_t)(uintptr_t)views)[__builtin_amdgcn_readfirstlane(i)];
#else
    return views[i];
#endif
}

// n / d for n < 2^31 without a division (Granlund & Montgomery): t = mulhi(n, m); q = (t + n) >> s, exact for every
// n below 2^31 (t + n cannot overflow there) with s = ceil(log2 d), m = floor(2^32 (2^s - d) / d) + 1 (host: fast_div).
struct FastDiv {
    uint32_t m, s;
};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return (__umulhi(n, f.m) + n) >> f.s; }

struct GridDesc {
    float ox, oy, oz, vs;
    uint32_t ny, nz;
    uint32_t i0;        // global x index of the engine's first plane
    uint32_t gpc;       // 4-voxel groups per row = nzp / 4 (the last ones of a padded row own fewer than 4 voxels, or none)
    uint64_t ngroups;   // columns owned * gpc
    uint32_t istride;   // global x step between the engine's planes (1: slab, W: plane-cyclic)
    uint32_t nzp;       // row pitch of the state in voxels: nz rounded up to a multiple of 64 (a row = one
                        // (plane, column) run of nz voxels, 256-byte aligned; the padding is never read back)
    FastDiv by_nzp, by_ny;  // for the survivor stages' decode of an entry (an element index below 2^31)
};

constexpr int kBlock = 256;
constexpr int kTile = 32;        // mask tile edge in pixels (32 rows x 32 bits = 128 B)
constexpr int kSub = 256;        // sharded append counters = sub-lists of a survivor list
constexpr int kStreamGroups = 2; // 16-byte groups per lane in the per-view streaming kernel
                                 // (measured with streaming loads: 2 -> 0.0803, 3 -> 0.0811, 4 -> 0.0860 ms)
constexpr int kXcdRun = 16;      // consecutive logical blocks kept on one XCD
constexpr int kCandSub = 8;      // sub-lists of the candidate list (see ListCtl::ncand)

// Survivor lists of the fused carve (see carve_list_kernel).  Zeroed before every fused launch.
// Every counter sits on a 128-byte line of its own: returning device-scope atomics on one
// line serialise (~90 per microsecond measured), on different lines they do not.
struct alignas(128) ListCounter {
    uint32_t n;
    // A reservation is a plain atomic add (a compare-and-swap loop measured 2 x slower on a scene that appends
    // 260 000 chunks).  One that does not fit leaves `n` beyond the capacity, and -- `n` only grows -- so does every
    // later one: the entries that were written are exactly those below the base of the FIRST reservation that failed.
    // That base is kept here (as its complement, so that zero means "none failed"; atomic max): readers stop there.
    uint32_t cut;
    uint32_t pad[30];
};
// entries of a sub-list a reader may trust: below its capacity and below the first failed reservation
__device__ __forceinline__ uint32_t list_count(const ListCounter &c, uint32_t cap) {
    return min(min(c.n, cap), c.cut ? ~c.cut : 0xffffffffu);
}
struct ListCtl {
    ListCounter count[5][kSub];  // entries appended per sub-list, one set per list stage; set 3: bulk units, set 4:
                                 // their work items
    uint32_t overflow;           // a sub-list ran out of room in the DENSE stage: the special kernel's dense pass takes over
    uint32_t nlive[2];           // brick form: bricks no view found empty (entries of the live list); the flags
                                 // kernel of launch q counts in word q & 1 and zeroes the other one, so launches
                                 // that keep the same block (fewer than 6 views: no survivor stages) need no memset
    uint32_t nlate;              // FULL candidates a later view did not keep whole (entries of the late list)
    uint32_t nfill[2];           // settled bricks that need a fill (entries of the fill list), same alternation
    uint32_t nlate_units;        // ... of them, those whose four units take the bulk units' road (round 5): entries at the
                                 // late list's far end, last one first (brick_confirm_kernel, carve_special_kernel)
    uint32_t pad[25];
    ListCounter xcd_next[64];    // dense stage: ticket counters for the live list, 8 per XCD (index xcd * 8 + c: the
                                 // wavefronts of XCD k whose number ends in c share one; see carve_brick_kernel)
    ListCounter ncand[2][kCandSub];  // FULL candidates the flags kernel left open for the confirm kernel: entries of the
                                 // candidate list's sub-lists, same alternation as nlive.  Sub-list s takes the blocks
                                 // [s per, (s + 1) per) of the flags kernel, one atomic per block with candidates --
                                 // inside a solid object that is every block at about the same time, and returning
                                 // atomics on one line serialise at 11 ns each: one counter for all of them (a 64-bit
                                 // add that reserved the live list's places as well) made the flags kernel 25 -> 35 us
                                 // there
};

// Bricks whose -1 fill is left to the final list stage (see carve_list_kernel).
struct CullStores {
    const uint8_t *flags;  // null: nothing deferred
    uint32_t bricks_y, bricks_z, nstrips, first;  // strips [first, nstrips) are filled there
    int32_t kept, fresh;   // see Fill
    uint32_t fill_blocks;  // 0: one store block per strip; n: n persistent store blocks
    int32_t init;          // see Fill
    uint32_t spec;         // strips [0, spec) were set to -1 ahead of the verdicts (SpecFill): this stage's store blocks
                           // also walk them, for their FULL / UNTOUCHED bricks only
};
