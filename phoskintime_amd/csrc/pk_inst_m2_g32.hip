#define PK_INST_MODEL 2
#define PK_INST_G 32
#include "pk_inst.inc"
