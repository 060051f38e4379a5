#pragma once
/* Put include/eigen_api in FRONT of include/ on the include path and the reference's `#include "ML/KMeans.hpp"` resolves to the
 * Eigen-typed, header-only API over the C handles (include/ML/EigenApi.hpp) instead of the C++ facade of include/ML/KMeans.hpp. */
#include "../../ML/EigenApi.hpp"
