#pragma once
#include "tbb_shim.h"
