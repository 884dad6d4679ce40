// fgm.hpp — fragment canvas (drop-in layer).
#pragma once
#include "../fragment.hpp"
