// nic.hpp — native image compression (drop-in layer).
#pragma once
#include "../nic.hpp"
