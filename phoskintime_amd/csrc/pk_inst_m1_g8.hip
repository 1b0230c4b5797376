#define PK_INST_MODEL 1
#define PK_INST_G 8
#include "pk_inst.inc"
