/*
 * spacecarve_tuning.h -- sc_set_option keys that only move work between the kernels of the fused carve /
 * averaging launch (defaults in parentheses).  RESULTS NEVER DEPEND ON THEM
 * (tests/test_parity_gpu.py::test_fused_pipeline_knobs_never_change_a_label, tools/fuzz_carve.py); they exist for
 * the sweeps behind DESIGN.md 4 and for tests that force a path.  The numbers are part of the ABI like the keys of
 * spacecarve.h (they share sc_set_option's key space) and are never renumbered.
 * Round 6 retired sixteen of them -- the ones the sweeps had settled (profiles/r05_sweep_defaults.json) -- to "accepted, no
 * effect": the fused carve runs with their defaults.  What remains are the switches that select a path (tests force
 * paths with them) and the few counts whose best value depends on the scene.
 */
#ifndef SPACECARVE_TUNING_H
#define SPACECARVE_TUNING_H

#define SC_OPT_COMPACT 5          /* carve only. 1 (default): a fused launch of >= 6 views is dense
                                     for its first two views, then finishes the survivors from
                                     compacted lists; 0: every view is applied densely           */

#define SC_OPT_DENSE_VIEWS 6      /* views applied to every voxel before compaction (2)       */
#define SC_OPT_STAGE1_VIEWS 7     /* views applied to the first survivor list (6; 8 until round 5) */
#define SC_OPT_LIST_BLOCKS 8      /* RETIRED (round 6: accepted, no effect; fixed at 2048).  Was: persistent grid of list stages without store blocks (2048)              */
#define SC_OPT_VIEW_GROUP 9       /* RETIRED (round 6: accepted, no effect; fixed at 2).  Was: the spans of the final survivor stage are a multiple of this many views (2) */
#define SC_OPT_BRICK 10           /* 1 (default): for grids with nz <= 4096 and < 2^31 voxels the dense stage
                                     works on 16x64-voxel bricks with a conservative emptiness
                                     test per brick; 0: linear blocks only                       */
#define SC_OPT_STAGE2_VIEWS 12    /* RETIRED (round 6: accepted, no effect; fixed at 0 (no such stage)).  Was: views applied to a second survivor list (0 = no such stage)  */
#define SC_OPT_PACK_ROWS 13       /* the mask bit packer: 0 (default) the band form -- a block takes one tile row of a
                                     view, whole picture rows read in a piece (pictures up to 2048 pixels wide, wider
                                     ones take the panel form) --; 1, 2, 4, 8: the panel form, 128-pixel panels of that
                                     many tile rows per block; 3: bands for every picture up to 2048 pixels wide     */
#define SC_OPT_DEFER_STORES 14    /* RETIRED (round 6: accepted, no effect; fixed at 1536).  Was: n > 0 (default 1536): the -1 fill of bricks found empty is done by
                                     store blocks running beside n persistent blocks of the final
                                     survivor stage; 0: by the dense stage                          */
#define SC_OPT_DEFER_SHARE 15     /* RETIRED (round 6: accepted, no effect; fixed at 16).  Was: sixteenths of the strips filled by the final stage (16); 0: none */
#define SC_OPT_FULL_BRICKS 19     /* 1 (default): a brick EVERY view of the batch sees whole, in-image, over
                                     foreground only gets its labels (0 -> 1) without projecting a voxel */
#define SC_OPT_AVG_BRICK 20       /* averaging. 1 (default): bricks whose footprint in a view is flat (all 0 / all 255 bytes,
                                     or one float32 value) add that view's value without projecting          */
#define SC_OPT_AVG_TILE_F32 29     /* averaging with float32 masks. 1 (default): the masks are re-laid in 8x4-pixel tiles
                                     (one 128-byte line each) with per-region uniformity, and take the brick
                                     form too (a footprint over ONE value adds it without projecting);
                                     0: gathered row-major as handed over                                  */
#define SC_OPT_STAGE1_STORE_SHARE 17 /* RETIRED (round 6: accepted, no effect; fixed at 5).  Was: sixteenths of those strips filled beside the FIRST survivor stage (5) */
#define SC_OPT_STAGE1_LIST_BLOCKS 21 /* RETIRED (round 6: accepted, no effect; fixed at 1280).  Was: persistent list blocks of that stage when it carries a share (1280)   */
#define SC_OPT_PACK_RIDE 22        /* 1 (default): a batch of device-resident 1-byte masks (sc_process_views_device)
                                     is packed when it is launched, in the order its views are applied: the
                                     first ones ahead, the rest beside the dense stage; 0: all at enqueue    */
#define SC_OPT_BRICK_WALKERS 23    /* RETIRED (round 6: accepted, no effect; fixed at 1280).  Was: persistent blocks of the dense stage when packing rides beside it (1280) */
#define SC_OPT_FILL_BLOCKS 25      /* RETIRED (round 6: accepted, no effect; fixed at 256).  Was: store blocks of a list stage: 0 one short block per strip of bricks, n > 0
                                     that many persistent blocks walking the strips (256 = one per CU: a
                                     wavefront's stores do not hold it up, so few keep the write path busy;
                                     512 until round 4) */
#define SC_OPT_FINAL_VOXELS 24      /* RETIRED (round 6: accepted, no effect; fixed at 2).  Was: survivors per lane in the final survivor stage: 1, 2 (default) or 4         */
#define SC_OPT_VIEW_BRICK 26        /* 1 (default): a launch of ONE view (the reference's cadence, cl.py:223-226)
                                     uses the brick verdicts too: bricks the view sees whole over background
                                     are carved blind and skipped by later views; 0: the streaming kernel
                                     (every view reads the whole state: the north star's formulation)          */
#define SC_OPT_STAGE1_VOXELS 30     /* RETIRED (round 6: accepted, no effect; fixed at 2).  Was: ... in the survivor stages before it: 1, 2 (default) or 4                  */
#define SC_OPT_FLAG_VIEWS 11      /* views of a batch that may declare a brick empty (8; 0 = all) */
#define SC_OPT_BULK_MIN 32        /* a wavefront's share of a live brick (a UNIT: 16 columns x 16 voxels) with at least this
                                     many voxels alive after the dense views is asked about as a whole: every remaining
                                     view at once, one view per lane, over 8x8-pixel cells of the masks; only the
                                     undecided views project its voxels (128; 0 = never)                          */
#define SC_OPT_ITEM_BIAS 33       /* RETIRED (round 6: accepted, no effect; fixed at 12).  Was: sixteenths (12): a unit's undecided views become work items of the final stage (half a
                                     unit x up to 16 views each) when those cost at most this share of what its voxels
                                     would cost in the survivor lists; otherwise the voxels take the lists        */
#define SC_OPT_UNIT_BLOCKS 34     /* RETIRED (round 6: accepted, no effect; fixed at 512).  Was: blocks of 8 wavefronts giving the units their verdicts (512)                  */
#define SC_OPT_BULK_FLOOR 35      /* bulk units a batch must have for their verdicts to be asked (8192 = two rounds of the
                                     special kernel's wavefronts): with fewer the verdict rounds are a latency chain
                                     nothing amortises (the bench's thin plant has 3 268 such units and their verdicts
                                     settle next to nothing; a bulky object 18 000 - 56 000) and the first survivor
                                     stage takes the units' voxels as they are.
                                     Decided on the device, inside the batch, from the count its own dense stage left:
                                     the first batch of an engine runs like every later one.  0: always asked        */
#define SC_OPT_BULK_ADAPT SC_OPT_BULK_FLOOR /* deprecated name of key 35 (rounds 3: 0 = always on, which 0 still means) */

#define SC_OPT_LIST_CAP 38        /* entries per survivor sub-list (0, the default: 5/16 of the voxels over the 256 sub-lists).
                                     A small value makes the lists overflow on a small grid: how the tests reach the
                                     overflow paths (the special kernel's dense pass, a bulk unit without room)      */
#define SC_OPT_HOST_PACK 39       /* 1 (default): a carve mask handed over in HOST memory (sc_process_view) is reduced to
                                     1 bit per pixel on host threads and crosses PCIe as bits (14 MB for 72 masks of
                                     1440 x 1080 instead of 112 MB), one copy per flush; 0: the bytes go through the
                                     page-locked ring and are packed on the device (round 1-3 path)                 */
#define SC_OPT_HOST_THREADS 40    /* threads of the library's host pool (mask bits, label widening): 0 = default,
                                     min(8, hardware threads / 2).  Process-wide; before the pool's first use        */
#define SC_OPT_LDS_TILES 41       /* accepted, no effect.  Was: the dense stage loads the window of mask words a unit's 256
                                     voxels project onto into LDS once per view and the voxels read their words from there
                                     -- the north star's "LDS-staged mask tiles per wavefront".  Built, exact, measured
                                     slower on every scene (DESIGN_APPENDIX.md 12: the gathers hit L2 and the stage is
                                     bound by its arithmetic) and removed when the stage's lane state moved to scalar masks */
#define SC_OPT_BULK_LIVE 42       /* sixteenths of the bricks (2): with fewer live bricks than that after the tile verdicts
                                     the batch is a thin object and no unit goes on the bulk list (decided by the dense
                                     stage on the device from the batch's own live count); 0: the list is always kept */
#define SC_OPT_SAFE_KERNELS 43    /* 1 (default): a batch whose views are ALL certified by the host (project(): every voxel
                                     of the grid well in front of the camera, everything finite -- any real rig) runs the
                                     survivor stages in instances compiled without the general path; 0: never        */
#define SC_OPT_DENSE_EXTRA 44     /* RETIRED (round 6: accepted, no effect; fixed at 1).  Was: 1 (default): a unit (16 columns x 16 voxels) that the dense views thinned out without
                                     emptying -- 32 .. 128 of its 256 voxels left -- takes one more pair of views inside the
                                     dense stage (masks that carve voxel by voxel: the noise scene); 0: never */
#define SC_OPT_SPEC_SHARE 45      /* RETIRED (round 6: accepted, no effect; fixed at 3).  Was: sixteenths of the strips (3) whose labels are set to -1 AHEAD of the brick verdicts, by
                                     persistent fill blocks in front of the flags kernel's own (fresh volumes only: every
                                     brick that is not EMPTY is written again by a later kernel of the batch); 0: none  */
#define SC_OPT_SPEC_BLOCKS 46     /* RETIRED (round 6: accepted, no effect; fixed at 64).  Was: ... that many blocks of 512 threads (64)                                             */
#define SC_OPT_LATE_ROAD 47       /* 1 (default): a brick every view packed ahead keeps whole and a later one does not (a
                                     "late" brick) gets its labels from the confirm kernel and its four units join the bulk
                                     units -- verdicts per unit, work items for the undecided views -- when the batch has a
                                     bulk list; 0: the late list (one wavefront takes a unit through every view), always  */
#define SC_OPT_UNIT_CULL 37       /* 1 (default): inside the dense stage the four units (16 columns x 16 voxels) of every
                                     live brick are asked about as a whole, over 8x8-pixel cells, by the views packed
                                     ahead; a unit some view finds empty is carved whole, not projected -- unless
                                     the tile level settled less than half of the bricks (masks without structure:
                                     nothing for the cells to find); 2: asked whatever the tiles settled; 0: never */

#endif /* SPACECARVE_TUNING_H */
