// ful.hpp — fragment files (drop-in layer).
#pragma once
#include "../ful.hpp"
