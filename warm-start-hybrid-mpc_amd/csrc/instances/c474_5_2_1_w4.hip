// One line of HMPC_INSTANCE_LIST (hmpc_kernel.hip): the cold and the hand-down kernel of this instantiation, compiled in a
// translation unit of its own so that the shipped build compiles eight of them at a time (csrc/Makefile).
#define HMPC_KERNEL_ONLY
#include "hmpc_device.h"
#include "hmpc_kernel.hip"
HMPC_INSTANCE(4, 7, 4, 5, 2, 1, 4)
