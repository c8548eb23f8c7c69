// goddard.cpp -- host side of the Goddard model mirror (reference: goddard.cpp:23-387).
#include "goddard.hpp"

#include <cmath>
#include <stdexcept>

#include "socp_hip.h"

struct goddard::data_struct {
    std::vector<real> switchingTimes;   // bang -> singular -> off instants (goddard.cpp:19)
};

namespace {
const char *const kParamOrder[SOCP_GODDARD_NPARAMS] = {"C", "b", "KD", "kr", "u_max", "mu1", "mu2", "singularControl"};
}

goddard::goddard(std::string the_fileTrace, int stepNbr) : model(7, 0, stepNbr, the_fileTrace), data(new data_struct)
{
    data->switchingTimes = {0.0227, 0.08};            // goddard.cpp:27-29
    const parameters_struct def;
    const real values[SOCP_GODDARD_NPARAMS] = {def.C, def.b, def.KD, def.kr, def.u_max, def.mu1, def.mu2, def.singularControl};
    for (int i = 0; i < SOCP_GODDARD_NPARAMS; i++) parameters[kParamOrder[i]] = values[i];
}

goddard::~goddard() { delete data; }

int goddard::DeviceModelId() const { return SOCP_MODEL_GODDARD; }

int goddard::DeviceParams(double *out, int cap) const
{
    if (cap < SOCP_GODDARD_NPARAMS) return 0;
    for (int i = 0; i < SOCP_GODDARD_NPARAMS; i++) out[i] = parameters.at(kParamOrder[i]);
    return SOCP_GODDARD_NPARAMS;
}

std::vector<real> goddard::DeviceSwitchingTimes() const { return data->switchingTimes; }

goddard::mstate goddard::Model(real const &t, mstate const &X, int) const { return DeviceEval(SOCP_EVAL_RHS, t, X, 0); }
goddard::mcontrol goddard::Control(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_CONTROL, t, X, 0); }
goddard::mstate goddard::Hamiltonian(real const &t, mstate const &X, int) const { return DeviceEval(SOCP_EVAL_HAMILTONIAN, t, X, 0); }

// goddard.cpp:298-317 is the generic segment integration; nothing model-specific to add
goddard::mstate goddard::ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac)
{
    return model::ModelInt(t0, X, tf, isTrace, isJac);
}

// goddard.cpp:320-340: the generic row plus the switching function mu1 - b p_m - C/m |p_v|
void goddard::Trace(real const &t, mstate const &X, std::stringstream &file) const
{
    const mstate u = Control(t, X);
    const mstate H = Hamiltonian(t, X, 0);
    file << t << "\t";
    for (int k = 0; k < 2 * dim; k++) file << X[k] << "\t";
    for (size_t k = 0; k < u.size(); k++) file << u[k] << "\t";
    file << H[0] << "\t";
    const real sw = parameters.at("mu1") - parameters.at("b") * X[13]
                    - parameters.at("C") / X[6] * std::sqrt(X[10] * X[10] + X[11] * X[11] + X[12] * X[12]);
    file << sw << std::endl;
}

// goddard.cpp:188-253: the closed-form singular thrust is only reachable through the control law
// on the device (the value of Control() on the singular arc divided out is not needed by callers)
real goddard::GetSingularControl(real, mstate const &) const
{
    throw std::logic_error("goddard::GetSingularControl: evaluated inside the device control law only");
}

// goddard.cpp:343-370: at a free interior time the row is H(t, X-)
goddard::mstate goddard::SwitchingTimesFunction(real const &t, mstate const &X, mstate const &, int isJac) const
{
    return Hamiltonian(t, X, isJac);
}

void goddard::SwitchingTimesUpdate(std::vector<real> const &switchingTimes) { data->switchingTimes = switchingTimes; }
void goddard::SetParameterDataName(std::string name, real value) { parameters.at(name) = value; }
real &goddard::GetParameterDataName(std::string name) { return parameters.at(name); }
