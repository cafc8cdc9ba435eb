// librife_hip: host side of the MI355X-native RIFE engine + the C-ABI declared in include/rife_hip.h.
//
// What of the reference this file replaces (all under /root/reference/src):
//   RIFE::RIFE / ~RIFE        rife.cpp:27-78     -> rife_hip_create / rife_hip_destroy
//   RIFE::load                rife.cpp:127-379   -> rife_hip_load  (ncnn::Net::load_param/load_model -> NcnnModel,
//                                                  pipeline creation -> kernels are compiled ahead of time)
//   RIFE::process_v4          rife.cpp:2462-3202 -> Engine::run_v4 (one fixed schedule instead of ncnn's graph walk)
//   ncnn VkCompute record/submit/wait (rife.cpp:2522-2530, 3176-3186) -> one HIP stream per in-flight pair
// There is deliberately no CPU path in this library: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <array>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// Three builds of these sources (csrc/Makefile): librife_hip.so = the PRODUCT (exports include/rife_hip.h, nothing else, one schedule);
// librife_hip_test.so (-DRIFE_HIP_TEST_BUILD) = + the parity taps / single-kernel entry points of include/rife_hip_test.h and the kernel-selection
// switches the kernel-vs-kernel tests flip; librife_hip_bench.so (-DRIFE_HIP_BENCH_BUILD) = + csrc/bench_hooks.h.
#if defined(RIFE_HIP_BENCH_BUILD) && !defined(RIFE_HIP_TEST_BUILD)
#define RIFE_HIP_TEST_BUILD 1
#endif
#include "../../include/rife_hip.h"
#ifdef RIFE_HIP_TEST_BUILD
#include "../../include/rife_hip_test.h"      // the parity taps and single-kernel entry points the test build also exports
#endif
#include "conv_mfma.h"
#include "conv_img.h"
#include "elementwise.h"
#include "elementwise_v2.h"
#include "stem_fused.h"
#include "stem_fused_v2.h"
#include "stem_rs.h"
#include "tail_rs.h"
#include "head_h2.h"
#include "conv_t64.h"
#include "conv_row.h"
#include "conv_rs.h"
#include "conv_rs2.h"
#ifdef RIFE_HIP_TEST_BUILD
#include "conv_ks.h"      // round-4 K-split trunk kernel: opt-in (RIFE_HIP_KS), measured slower with pairs in flight; not compiled into the product
#endif
#include "graph_kernels.h"
#include "model_hashes.h"
#include "ncnn_model.h"

#include "engine_switches.h"
#include "engine_layers.h"
#include "engine_dispatch.h"
#include "engine_ctx.h"
#include "engine_v4.h"
#include "engine_v2.h"
#include "engine_v1.h"
#include "engine_abi.h"
