// temporary: replaced by vo_match.hip / vo_geometry.hip
#include "vo_common.hpp"
int vo_geometry_init(vo_ctx*) { return VO_OK; }
void vo_pairset_free(PairSet&) {}
