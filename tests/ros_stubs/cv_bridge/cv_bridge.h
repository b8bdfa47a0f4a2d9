// test stub: see ../ros_stub_core.h (tests/ros_stubs)
#pragma once
#include "ros_stub_core.h"
