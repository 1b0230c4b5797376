#define PK_INST_MODEL 0
#define PK_INST_G 8
#include "pk_inst.inc"
