/*
 * socp_plugin.h -- out-of-tree device models.
 *
 * The reference's plugin surface is a host C++ virtual (odeTools.hpp:82 Model, model.hpp:375 Control,
 * model.hpp:384 Hamiltonian); a virtual on the host cannot run inside a GPU kernel, so a model needs a
 * device twin.  In-tree models (goddard, doubleIntegrator, covid19) ship theirs; any other model class
 * provides one as a small plugin: a struct with the static device interface described in
 * socp_amd/csrc/plugin_impl.hpp, compiled by hipcc against this library's headers into a shared object.
 * The C++ mirror class of that model returns the plugin's id from model::DeviceModelId().
 *
 * Model ids below SOCP_PLUGIN_ID_MIN are reserved for in-tree models.
 */
#ifndef SOCP_PLUGIN_H_
#define SOCP_PLUGIN_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SOCP_PLUGIN_ID_MIN 100

/* dlopen `path`, call its socp_plugin_register(); returns SOCP_OK or a negative SOCP_ERR_* */
int socp_plugin_load(const char *path);

/* called BY the plugin: `table` is a socp::ModelLaunchers (launch.hpp) of `table_bytes` bytes */
int socp_register_model(int model_id, const void *table, int table_bytes);

/* every plugin exports this */
int socp_plugin_register(void);

#ifdef __cplusplus
}
#endif
#endif
