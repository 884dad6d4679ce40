// kpr.hpp — keypoint representation (drop-in layer).
#pragma once
#include "../grid.hpp"
