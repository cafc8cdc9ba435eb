// Structural hashes (NcnnModel::structural_hash) of the named output blobs of the graphs the compiled-in
// schedules were written for.  Regenerate with tools/param_hash.py after changing the hash function.
#pragma once
#define RIFE_V46_HASH_OUT0 0xee408936024d43cfull   /* models/rife-v4.6/flownet.param, blob "out0" */
#define RIFE_V40_HASH_OUT0 0xc679a7939b863e18ull   /* models/rife-v4/flownet.param, blob "out0" */
/* models/rife-v2.3 (== rife-v2, rife-v2.4): flownet "flow", contextnet "f1".."f4", fusionnet "output" */
#define RIFE_V23_HASH_FLOW 0xaf09294daee7aff7ull
#define RIFE_V3_HASH_FLOW 0x2f652fdad242a6faull    /* models/rife-v3.0, rife-v3.1 flownet "flow" (contextnet / fusionnet hash like v2.3) */
#define RIFE_V23_HASH_F1 0x91ca51f8d25c3b93ull
#define RIFE_V23_HASH_F2 0x1e6a1b4dc31dd611ull
#define RIFE_V23_HASH_F3 0x93dd7f70618876aeull
#define RIFE_V23_HASH_F4 0xcd053a14ca3d51b3ull
#define RIFE_V23_HASH_OUTPUT 0x35232d8b3d9a88a2ull
