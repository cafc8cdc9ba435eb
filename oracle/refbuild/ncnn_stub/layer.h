// ncnn look-alike header for oracle/refbuild (see ncnn_stub.h): the reference includes "layer.h"
#include "ncnn_stub.h"
