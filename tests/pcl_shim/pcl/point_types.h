#include <pcl/pcl_shim_core.h>
