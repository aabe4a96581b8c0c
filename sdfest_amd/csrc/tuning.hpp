// tuning.hpp -- every build-time tunable of libsdfr_hip.so in one place (gfx950).
// These are NUMBERS, not code paths: tile geometry, table sizes, thresholds between launch shapes.  The timing
// harness (tools/microbench/build_variant.sh NAME -DSDFR_...=...) builds variants by overriding them;
// tests/test_build_variants_cpu.py compiles a build with every one of them changed.  Experiments that needed other
// CODE (two rays per lane, Morton lane order, persistent grids, ...) are closed and removed; their numbers are in
// DESIGN.md sections 8 and 9 and profiles/r0*_experiments.txt.
#pragma once

// a workgroup's tile = SX x SY sub-tiles of 32 x 8 pixels: batch tile of both image kernels (64 x 8) ...
#ifndef SDFR_MACRO_SX
#define SDFR_MACRO_SX 2
#define SDFR_MACRO_SY 1
#endif
// ... and the forward's own batch tile (defaults to the same)
#ifndef SDFR_FWD_SX
#define SDFR_FWD_SX SDFR_MACRO_SX
#define SDFR_FWD_SY SDFR_MACRO_SY
#endif
// calls with at least this many 64 x 8 tiles take the batch tile, smaller ones the 32 x 8 sub-tile
#ifndef SDFR_BWD_MACRO_MIN
#define SDFR_BWD_MACRO_MIN 16384
#endif
#ifndef SDFR_FWD_MACRO_MIN
#define SDFR_FWD_MACRO_MIN 16384
#endif
// 1: medium-sized batches of wide images take 128 x 8-pixel forward tiles walked by 4 waves (common.hpp)
#ifndef SDFR_FWD_WIDE
#define SDFR_FWD_WIDE 1
#endif
// waves per workgroup of the batch forward (64 x 8 tiles)
#ifndef SDFR_FWD_WAVES
#define SDFR_FWD_WAVES 2
#endif
// pixels per voxel from which a view's backward tiles are 32 x 32 instead of 64 x 8
#ifndef SDFR_BWD_BIG_MIN_RATIO
#define SDFR_BWD_BIG_MIN_RATIO 2.0f
#endif
// face records are packed for a grid shared by at least this many views.  Packing costs the prologue ~4.4 us and 4 MiB
// of L2 (the launches behind it run slower), and pays by the march samples it serves: for objects that fill the
// screen (the benchmark's) from ~10 views on -- stand-alone forward of 12 / 16 / 24 / 32 views packed 30.7 / 33.3 /
// 42.2 / 48.1 us, plain grid 36.1 / 39.3 / 54.9 / 76.4 --, for mug-sized objects (the C5 loop: short marches) only
// from ~32: ms per iteration of the V-view loop, packed from 7 / 17 / 33 views on: V = 8 0.150 / 0.139 / 0.139,
// V = 16 0.169 / 0.156 / 0.156, V = 24 0.196 / 0.196 / 0.182, V = 32 0.190 / 0.190 / 0.185.  17: the loop's gain, and
// at most +6 us for a 16-view forward of screen-filling objects.  (4 until round 5.)
#ifndef SDFR_PACKED_MIN_VIEWS
#define SDFR_PACKED_MIN_VIEWS 17
#endif
// slots of the batch backward's z-pair run table (the fall-back of the dense box)
#ifndef SDFR_BWD_SLOTS
#define SDFR_BWD_SLOTS 512
#endif
// words of the backward's dense LDS box: 18 KiB, the size of the batch hash it shares the LDS with
#ifndef SDFR_DENSE_CAP
#define SDFR_DENSE_CAP 4608
#endif
// most views the loop's one-launch render step takes (every workgroup derives its view's record itself; every view
// has its own unscaled d/dSDF volume).  The tail form of the loop ends at 7 views.
#ifndef SDFR_FUSED_MAX_VIEWS
#define SDFR_FUSED_MAX_VIEWS 8
#endif
// the one-launch render step sends a view's d/dSDF straight to float atomics while the view's observed point set (its
// mask's pixels) is at most this large, and through the LDS tables beyond (tools/microbench/fused_render_close.py: the
// C5 mug at 0.5 m, 4 k pixels: straight 0.0995 ms per iteration against 0.1042 through the tables; 11 k pixels 0.117
// against the two launches' 0.112; 25 k: 0.145 against 0.119; 83 k: 0.306 against 0.135)
#ifndef SDFR_FUSED_DIRECT_MAX_POINTS
#define SDFR_FUSED_DIRECT_MAX_POINTS 6144
#endif
// 1: the two-launch form's small-tile backward and its sampler blocks send d/dSDF straight to the volume's float
// atomics, as the one-launch step does for small objects (render_fused_l1_pc_kernel), instead of pre-summing in LDS.  Measured on
// the C5 scene seen from V cameras, ms per iteration (tools/microbench/loop_forms.py): 2 views 0.1125 -> 0.1089, 4 views
// 0.1207 -> 0.1233, 8: 0.1330 -> 0.1499, 16: 0.1624 -> 0.1979; K objects side by side unchanged -- beyond a view or two
// the atomics on a shared volume collide (and a stand-alone backward does not know how large its object is: BASELINE's C1,
// 160x120, 26.3 -> 19.4 us per pair, but C2, 640x480 and a large object, 23.0 -> 33.1), so: 0
#ifndef SDFR_SMALL_DIRECT
#define SDFR_SMALL_DIRECT 0
#endif
// sampler side of sdfr_render_*backward_l1_pc: workgroups the launch aims at over all views, and the least a view gets
// (sampler_device.hpp, GROUPS)
#ifndef SDFR_PC_GRID_TARGET
#define SDFR_PC_GRID_TARGET 8192
#endif
#ifndef SDFR_PC_MIN_GROUPS
#define SDFR_PC_MIN_GROUPS 64
#endif
// batched decoder: the direct (VALU) convolution from this many workgroups (tiles x latents) on, the MFMA form below
#ifndef SDFR_DIRECT_MIN_TILES
#define SDFR_DIRECT_MIN_TILES 512
#endif
// ... and for SDFR_DIRECT_MIN_LATENTS_FEW or more latents on layers of at least 15 tiles
#ifndef SDFR_DIRECT_MIN_TILES_FEW
#define SDFR_DIRECT_MIN_TILES_FEW 240
#endif
#ifndef SDFR_DIRECT_MIN_LATENTS_FEW
#define SDFR_DIRECT_MIN_LATENTS_FEW 8
#endif
// decoder forward: the last resize with the 1x1x1 layer inside it (one gather kernel, eight gathers per output) up to
// 2^this output elements -- one or two mug latents -- conv1x1 + the tiled resize above (bit-identical; at 2^21 the
// 4- and 8-object loops ran 431 / 680 objects/s, at 2^19 446 / 744)
#ifndef SDFR_FEW_MIX_LOG2
#define SDFR_FEW_MIX_LOG2 19
#endif
// decoder forward: an up-sampling resize takes the LDS-tiled kernel from this many (tile, volume) items on, the gather
// kernel below (a single decode -- 128 and 64 items at the mug decoder's two up-sampling resizes -- has too few tiles to
// fill the chip; from two latents on the tiled kernel wins: at 2048 the 4- / 8- / 16-object loops ran 474 / 778 / 1263
// objects/s, at 256 489 / 840 / 1360)
#ifndef SDFR_RESIZE_TILED_MIN_ITEMS
#define SDFR_RESIZE_TILED_MIN_ITEMS 256
#endif
// a step's forward over at most this many views of the plain grid sets its views up itself (no prologue launch): the
// inline form's threads also zero-fill the step's gradient volume(s), which grows with the views -- ms per iteration of
// the V-view loop at 6 / 4 / 3 / 2: V = 4 0.1305 / 0.1315 / 0.1288 / 0.1283, V = 6 0.1384 / 0.1344 / - / -; K objects
// with their own grids, objects/s: K = 4 488 / 485 / 495 / 495, K = 6 641 / 655 / - / -, K = 8 (at 8) 839 -> 864
#ifndef SDFR_INLINE_MAX_VIEWS
#define SDFR_INLINE_MAX_VIEWS 3
#endif
// decoder: the split-K form of the MFMA convolution (four waves share a tile) up to this many latents -- small
// batches then equal single decodes bit for bit (64 measured: the 32- and 64-object loops 1 915 -> 1 863, 2 433 -> 2 227
// objects/s)
#ifndef SDFR_SPLITK_MAX_LATENTS
#define SDFR_SPLITK_MAX_LATENTS 16
#endif
// waves per SIMD the backward kernel's register allocation is held to (0: the compiler's choice)
#ifndef SDFR_BWD_WAVES_PER_EU
#define SDFR_BWD_WAVES_PER_EU 8
#endif
// decoder VJP, resize3_backward_tiled_kernel: 16-byte vectors a thread prefetches per channel, elements it stores per
// channel (together ~100 VGPRs at 4 / 3: 1024 threads per workgroup), and tile edge / workgroup size forced for
// timing (0: picked per resize from LDS and register occupancy)
#ifndef SDFR_BT_LOADS
#define SDFR_BT_LOADS 4
#endif
#ifndef SDFR_BT_OUT
#define SDFR_BT_OUT 3
#endif
#ifndef SDFR_BT_TILE
#define SDFR_BT_TILE 0
#endif
#ifndef SDFR_BT_THREADS
#define SDFR_BT_THREADS 0
#endif
