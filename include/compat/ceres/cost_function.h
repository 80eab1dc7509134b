#pragma once
#include "ceres.h"
