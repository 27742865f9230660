#ifndef SSG_K_SMEM2_H
#define SSG_K_SMEM2_H
#include "ssg_dev.h"
#endif
