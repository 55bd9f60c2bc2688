/* tests/r_mock (see Rinternals.h) */
#include "Rinternals.h"
