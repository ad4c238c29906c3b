// probe.hip.h -- umbrella header of the device code of the interval join (see index_view.hip.h for
// the HBM layout and the predicate).
#pragma once
#include "index_view.hip.h"
#include "index_build.hip.h"
#include "overlap.hip.h"
#include "count_nearest.hip.h"
#include "partition.hip.h"
#include "flat.hip.h"
#include "materialize.hip.h"
#include "sortscan.hip.h"
