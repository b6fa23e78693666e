// grid_mapf.hpp -- see grid_mapf.cpp; the C ABI is in include/dlscqp.h (dlscqp_mapf_*).
#pragma once
#include "../../include/dlscqp.h"
