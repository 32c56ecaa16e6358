#include "../pcl_shim_core.h"
