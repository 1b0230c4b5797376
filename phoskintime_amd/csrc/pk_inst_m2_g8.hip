#define PK_INST_MODEL 2
#define PK_INST_G 8
#include "pk_inst.inc"
