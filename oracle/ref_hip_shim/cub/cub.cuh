#pragma once
// cub:: -> hipcub:: (same DeviceScan::InclusiveSum / DeviceRadixSort::SortPairs signatures, rocPRIM underneath)
#include <cstring>
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
