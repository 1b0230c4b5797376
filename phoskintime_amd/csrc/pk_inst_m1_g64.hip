#define PK_INST_MODEL 1
#define PK_INST_G 64
#include "pk_inst.inc"
