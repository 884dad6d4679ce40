// kpm.hpp — device-backed drop-in (see remap/collector.hpp).
#pragma once
#include "../collector.hpp"
