/*
 * gmg_icm.h -- C ABI over the host-side ICM_t (glimmer-mg_amd/host/icm.hh) for
 * FFI users that cannot include the C++ class.  Model I/O and the null-model
 * builder are host code, as in the reference (src/ICM/icm.cc:65-216, 614-803);
 * nothing here scores (gmg_icm_train counts on the device).  Status codes and gmg_last_error() as in gmg.h; no
 * function exits the process.
 */
#ifndef GMG_ICM_H
#define GMG_ICM_H

#include "gmg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmg_icm gmg_icm;   /* owns one ICM_t */

/* ICM_t::ICM_t(m, d, p)                          src/ICM/icm.cc:24-44   */
int gmg_icm_new(int model_len, int model_depth, int periodicity, gmg_icm **out);
/* ICM_t::Read / Input without the exit()         src/ICM/icm.cc:614-726,846-861 */
int gmg_icm_open(const char *path, gmg_icm **out);
/* ICM_t::Build_Indep_WO_Stops; the model must be (3,2,3)  src/ICM/icm.cc:65-216 */
int gmg_icm_build_indep(gmg_icm *icm, double gc_frac, const char *const *stop_codon, int n_stops);
/* ICM_t::Output(fp, binary)                      src/ICM/icm.cc:729-803,961-998 */
int gmg_icm_write(gmg_icm *icm, const char *path);
int gmg_icm_free(gmg_icm *icm);
/* ICM_Training_t(m, d, p) + Train_Model(data)   src/ICM/icm.cc:1010-1042,1356-1455: a model trained on n_strings
 * NUL-terminated lower-case strings (build-icm's Training_Data).  The counting runs on the device (gmg_trainer_*). */
int gmg_icm_train(const char *const *strings, int n_strings, int model_len, int model_depth, int periodicity,
                  gmg_icm **out);

int gmg_icm_params(const gmg_icm *icm, int *model_len, int *model_depth, int *periodicity, int *num_nodes);
/* copies mip[P*N] and prob4[P*N*4] (the layout gmg_model_upload takes) */
int gmg_icm_tables(const gmg_icm *icm, int16_t *mip, float *prob4);
/* device mirror of the tables: uploaded on first use, owned by the gmg_icm */
int gmg_icm_device_model(const gmg_icm *icm, const gmg_model **out);

#ifdef __cplusplus
}
#endif
#endif
