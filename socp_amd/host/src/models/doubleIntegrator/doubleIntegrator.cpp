// doubleIntegrator.cpp -- host side of the double integrator mirror (reference:
// doubleIntegrator.cpp:26-315).
#include "doubleIntegrator.hpp"

#include "socp_hip.h"

struct doubleIntegrator::data_struct {
    doubleIntegrator::parameters_struct parameters;
    int stepNbr;
};

doubleIntegrator::doubleIntegrator(int modelOrder, std::string the_fileTrace)
    : model(6, modelOrder, 30, the_fileTrace), data(new data_struct), my_odeStruct(nullptr)
{
    data->parameters.u_max = 1;       // doubleIntegrator.cpp:30-32
    data->parameters.a_max = 1;
    data->parameters.muT = 0.01;
    data->stepNbr = 0;
}

doubleIntegrator::~doubleIntegrator() { delete data; }

doubleIntegrator::parameters_struct &doubleIntegrator::GetParameterData() { return data->parameters; }
void doubleIntegrator::SetStepNumber(int step) { data->stepNbr = step; }

int doubleIntegrator::DeviceModelId() const { return SOCP_MODEL_DOUBLE_INTEGRATOR; }

int doubleIntegrator::DeviceParams(double *out, int cap) const
{
    if (cap < SOCP_DINT_NPARAMS) return 0;
    out[0] = data->parameters.u_max;
    out[1] = data->parameters.a_max;
    out[2] = data->parameters.muT;
    return SOCP_DINT_NPARAMS;
}

doubleIntegrator::mstate doubleIntegrator::Model(real const &t, mstate const &X, int isJac) const
{
    return isJac == 0 ? ModelState(t, X) : ModelJacobian(t, X);
}
doubleIntegrator::mstate doubleIntegrator::ModelState(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_RHS, t, X, 0); }
doubleIntegrator::mstate doubleIntegrator::ModelJacobian(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_RHS, t, X, 1); }
doubleIntegrator::mcontrol doubleIntegrator::Control(real const &t, mstate const &X) const { return DeviceEval(SOCP_EVAL_CONTROL, t, X, 0); }
doubleIntegrator::mstate doubleIntegrator::Hamiltonian(real const &t, mstate const &X, int isJac) const { return DeviceEval(SOCP_EVAL_HAMILTONIAN, t, X, isJac); }
