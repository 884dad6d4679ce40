// mpb.hpp — map building: the collect and splice stages (drop-in layer).
#pragma once
#include "../splice.hpp"
