/*
 * ssg_seed.cpp -- the seeding stage's kernels in a translation unit of their own (bwt_extend, bwt_smem1a,
 * bwt_seed_strategy1, mem_collect_intv).  A small unit compiles in seconds, so variants of the kernel can be built side by side and
 * checked on the MI355X in one GPU call (`make variant VUNITS=ssg_seed`, tools/dbg/seed_variants.sh).
 */
#include <algorithm>
#include "ssg_rt.h"
#include "k_smem2.h"
#include "../../include/ssgpu.h"
#include "ssg_index_int.h"

SSG_ABI_FP_DEFINE(seed)
