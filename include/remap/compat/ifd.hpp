// ifd.hpp — forwards to the consolidated type header of the drop-in layer.
#pragma once
#include "../core.hpp"
