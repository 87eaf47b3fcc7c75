// dst_gather.cpp — multi-GPU result gather behind the C ABI (filled in below).
#include "dst_internal.h"
