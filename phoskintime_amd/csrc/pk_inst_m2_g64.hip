#define PK_INST_MODEL 2
#define PK_INST_G 64
#include "pk_inst.inc"
