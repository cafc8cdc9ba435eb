// Structural hashes (NcnnModel::structural_hash) of the named output blobs of the graphs the compiled-in
// schedules were written for.  Regenerate with tools/param_hash.py after changing the hash function.
#pragma once
#define RIFE_V46_HASH_OUT0 0xee408936024d43cfull   /* models/rife-v4.6/flownet.param, blob "out0" */
