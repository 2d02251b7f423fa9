/* placeholder translation unit until the object rows land */
int orc_oracle_objects_placeholder(void) { return 0; }
