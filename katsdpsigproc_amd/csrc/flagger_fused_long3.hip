// Fused flagger for 8193 .. 12288 channels (three groups of runs per lane): the same
// source as flagger_fused_long.hip, compiled separately so that the two build in parallel.
#define KSP_LONG_THREE_GROUPS
#include "flagger_fused_long.hip"
