// sdp_window.hpp -- F1/F2 LDS-window period kernel (filled in after the generic path is green).
#pragma once
#include "sdp_device.hpp"
