#include "mex.h"
