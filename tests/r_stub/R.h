#include "Rinternals.h"
