// The opaque context of the C ABI (include/mugd.h), shared by api.hip and train.hip.
#pragma once
#include "net.h"

struct mugd_ctx {
    Ctx c;
};
