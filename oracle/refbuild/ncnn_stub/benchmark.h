// ncnn look-alike header for oracle/refbuild (see ncnn_stub.h): the reference includes "benchmark.h"
#include "ncnn_stub.h"
