// fgs.hpp — fragment splicing (drop-in layer).
#pragma once
#include "../splice.hpp"
