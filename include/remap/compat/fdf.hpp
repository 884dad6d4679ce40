// fdf.hpp — foreground filtering (drop-in layer).
#pragma once
#include "../fdf.hpp"
