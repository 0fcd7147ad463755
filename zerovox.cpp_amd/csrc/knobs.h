// knobs.h — the library's test / measurement switches in ONE table.
//
// None is needed in production.  Tests force a kernel regime (the regimes give the same bits), measurements A/B a tile
// shape.  The table changes only through the C-ABI test entry zv_debug_set(name, value) — the shipped library never reads
// the environment (the Python test / bench binding forwards ZV_* variables through that entry, so a shell script can still
// A/B a run); a diagnostic build (-DZV_DIAG) also fills it from the environment when it is loaded.
#pragma once

namespace zv
{

// X(name, built-in default): the enum and the table are generated from this one list, so their orders cannot drift apart
#define ZV_KNOB_LIST(X)                                                                                                               \
    X(ZV_NO_FUSE, 0)           /* 1: one launch per conv of a residual block */                                                       \
    X(ZV_NO_TRIPLE, 0)         /* 1: never the whole-block kernels */                                                                 \
    X(ZV_FUSE256, 0)           /* 1: fused pair kernel for the 256-channel stage at any length */                                     \
    X(ZV_NO_MERGE, 0)          /* 1: three branch outputs instead of their sum */                                                     \
    X(ZV_MERGE_ALWAYS, 0)      /* 1: the merged MRF sum at any length (default: only with rounds of workgroups to spare) */           \
    X(ZV_MERGE_SEQ, 1)         /* 0: the 256-channel stage's MRF sum by one three-branch workgroup per tile instead of three launches */ \
    X(ZV_MERGE_MAXC, 256)      /* widest stage whose last dilation pair stores the merged sum (batches) */                            \
    X(ZV_TAIL_GROUPS, 8)       /* utterance groups of a batch's last vocoder stage */                                                 \
    X(ZV_ARENA_FILL, 0)        /* byte a fresh activation arena is filled with (255: NaN patterns) */                                 \
    X(ZV_DEC_PREPASS, -1)      /* -1 auto, 0 decoder convs normalise on the fly, 1 f16 operand pass */                                \
    X(ZV_CONV_MT, 0)           /* minimum tile height of the generic conv kernel */                                                   \
    X(ZV_CONV_NT, 0)           /* 1 / 2: output tiles per wave of the generic conv kernel */                                          \
    X(ZV_CONV_SINGLE, 1)       /* 0: never the single-utterance MFMA loop, 2: also for one-chunk convs */                             \
    X(ZV_CONV_GEMM, 1)         /* 0 never, 1 batches, 2 always: conv_gemm_kernel for wide convs over an f16 operand tensor */         \
    X(ZV_CONV_STREAM, 1)       /* 0 never, 1 batches, 2 always: memory-bound 3-tap convs (the last upsample convs) on conv_stream_kernel */ \
    X(ZV_UP_GEMM, 1)           /* 0 never, 1 batches, 2 always: the wide upsample convs behind an f16 operand pass on conv_gemm_kernel */ \
    X(ZV_GEMM_ORDER, 2)        /* conv_gemm_kernel's workgroup order: 0 plain (group fastest), 2 the 9-tile group first */ \
    X(ZV_PAIR_MT, 0)           /* 2 / 3 / 4: tile height of the pair kernels */                                                       \
    X(ZV_CONV_XCD, 1)          /* single-utterance conv launches: a channel group's row tiles all on one XCD (its L2 holds the group's weights); 0 = row tiles dealt over the XCDs */ \
    X(ZV_CONV_WARM, 1)         /* single-utterance convs: a channel group's row tiles touch the group's weights (one load per 128-byte line) before they start: L2 hits instead of a miss shared by all of them per fragment */ \
    X(ZV_LINEAR_MERGED, 1)     /* the encoder's per-token layers of a batch (linear, 1-tap conv, plain LayerNorm) over all token rows as one dense segment (0: row tiles per utterance) */ \
    X(ZV_BLOCK64, 3)           /* 64-channel stage of a batch: branches with at most that many taps run their first two dilation pairs in one launch (resblock_block64_kernel); 0 never, negative: at any length */ \
    X(ZV_PAIR64_RING, 1)       /* 0 never, 1 batches, 2 always: 64-channel pair kernel with the weights through an LDS ring */        \
    X(ZV_TRIPLE_V2, 1)         /* 0 never, 1 batches, 2 always, 3 always on 512-row tiles: whole-block kernel with its weights in LDS */                      \
    X(ZV_TRIPLE_DB, 1)         /* 0: one weight buffer for every branch */                                                            \
    X(ZV_TRIPLE_INTERLEAVE, 1) /* 0: branches not interleaved per XCD */                                                              \
    X(ZV_ATT_SCALAR, 0)        /* 1: scalar attention kernel */                                                                       \
    X(ZV_ATT_MFMA, 0)          /* 1: matrix-core attention kernel whatever the size */                                                \
    X(ZV_LN_TAIL, 1)           /* 0: the style add, the predictors' linear layer and the bucket + embedding step as launches of their own instead of tails of a LayerNorm launch */

// Diagnostic builds only (-DZV_DIAG: scripts/stamps*.py, ablation timings): switches that produce WRONG results or change
// occupancy on purpose.  The shipped library does not contain them, nor the code they select (kernels.h: ZV_DBGBITS).
#ifdef ZV_DIAG
#define ZV_KNOB_LIST_DIAG(X)                                                                                                          \
    X(ZV_DBG, 0)               /* timing-only ablation bits (wrong results) */                                                        \
    X(ZV_LDS_PAD, 0)           /* diagnostic: extra bytes of LDS per workgroup of the pair kernels (a lower occupancy on purpose) */ \
    X(ZV_STAMP_CP, 0)          /* diagnostic build: channel count of the pair launches that write phase stamps */                     \
    X(ZV_STAMP_CONV, 0)        /* diagnostic build: grid.y of the conv launches that write phase stamps */                            \
    X(ZV_STAMP_CIN, 0)         /* diagnostic build: their input channels */
#else
#define ZV_KNOB_LIST_DIAG(X)
#endif

enum Knob : int
{
#define ZV_KNOB_ENUM(n, d) n,
    ZV_KNOB_LIST(ZV_KNOB_ENUM)
    ZV_KNOB_LIST_DIAG(ZV_KNOB_ENUM)
#undef ZV_KNOB_ENUM
    ZV_KNOB_COUNT
};

int         knob(Knob k);
// false when no knob has that name
bool        knob_set(const char *name, int value);
// every knob back to its built-in default
void        knob_reset();
// false when no knob has that name
bool        knob_get(const char *name, int *value);
// bumped by every knob_set / knob_reset: captured graphs are keyed on it (a graph replays the regime it was captured in)
unsigned    knob_epoch();
const char *knob_name(int k);
// the timing-only ablation bits handed to the kernels: ZV_DBG in a diagnostic build, always 0 in the shipped library
#ifdef ZV_DIAG
inline int diag_bits() { return knob(ZV_DBG); }
#else
inline int diag_bits() { return 0; }
#endif

}  // namespace zv
