void R_CheckUserInterrupt(void);
