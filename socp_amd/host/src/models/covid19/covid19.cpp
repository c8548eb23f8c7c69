// covid19.cpp -- host side of the SEIR model mirror (reference: covid19.cpp:17-195).
#include "covid19.hpp"

#include "socp_hip.h"

struct covid19::data_struct {
    covid19::parameters_struct parameters;
    int stepNbr;                       // the model integrates with its own step count (covid19.cpp:36)
    std::string strFileTrace;
};

covid19::covid19(std::string the_fileTrace) : model(4), data(new data_struct)
{
    const parameters_struct def = {4, 10, 5, 1, 0.1, 1, -10, 20};      // covid19.cpp:28-35
    data->parameters = def;
    data->stepNbr = 1000;
    data->strFileTrace = the_fileTrace;
    strFileTrace = the_fileTrace;      // the base constructor got the default "": keep both in step
    std::ofstream wipe(data->strFileTrace.c_str(), std::ios::trunc);
}

covid19::~covid19() { delete data; }
covid19::parameters_struct &covid19::GetParameterData() { return data->parameters; }

int covid19::DeviceModelId() const { return SOCP_MODEL_COVID19; }
int covid19::DeviceStepNumber() const { return data->stepNbr; }

int covid19::DeviceParams(double *out, int cap) const
{
    if (cap < SOCP_COVID_NPARAMS) return 0;
    const parameters_struct &p = data->parameters;
    const double v[SOCP_COVID_NPARAMS] = {p.R0, p.Tinf, p.Tinc, p.N, p.Imax, p.muI, p.umin, p.umax};
    for (int i = 0; i < SOCP_COVID_NPARAMS; i++) out[i] = v[i];
    return SOCP_COVID_NPARAMS;
}

covid19::mstate covid19::Model(real const &t, mstate const &X, int) const { return DeviceEval(SOCP_EVAL_RHS, t, X, 0); }
covid19::mcontrol covid19::Control(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_CONTROL, t, X, 0); }
covid19::mstate covid19::Hamiltonian(real const &t, mstate const &X, int) const { return DeviceEval(SOCP_EVAL_HAMILTONIAN, t, X, 0); }

// covid19.cpp:167-190: the generic segment integration with dt = (tf - t0)/data->stepNbr
covid19::mstate covid19::ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac)
{
    const real dt = (tf - t0) / data->stepNbr;
    mstate Xs = X;
    if (isTrace) {
        std::stringstream ss;
        integrate(modelStruct(this, isJac), Xs, t0, tf, dt, observerStruct(this, ss));
        std::ofstream fileTrace(data->strFileTrace.c_str(), std::ios::app);
        fileTrace << ss.str();
    } else {
        integrate(modelStruct(this, isJac), Xs, t0, tf, dt);
    }
    return Xs;
}
