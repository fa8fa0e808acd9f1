// ivf.h — IVF-Flat kernels (k-means assignment/update, slab scan).  See ivf_host.inc.
#pragma once
#include "kernels.h"
