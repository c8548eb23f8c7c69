// doubleIntegrator.hpp -- host mirror of the reference's 3-D double integrator model class
// (doubleIntegrator.hpp:15-81); dynamics on the device (SOCP_MODEL_DOUBLE_INTEGRATOR).
#ifndef SOCP_AMD_DOUBLEINTEGRATOR_HPP_
#define SOCP_AMD_DOUBLEINTEGRATOR_HPP_

#include "../../socp/model.hpp"

#include <iostream>

class doubleIntegrator : public model
{
public:
    struct parameters_struct {
        real u_max;   // control bound
        real a_max;   // acceleration scale
        real muT;     // weight of time in the cost
    };

    doubleIntegrator(int modelOrder, std::string the_fileTrace);
    virtual ~doubleIntegrator();

    struct odeStruct;
    odeStruct *my_odeStruct;

    parameters_struct &GetParameterData();
    void SetStepNumber(int step);   // stored, never read -- as in the reference (doubleIntegrator.cpp:313-315)

    virtual int DeviceModelId() const;
    virtual int DeviceParams(double *out, int cap) const;

private:
    struct data_struct;
    data_struct *data;

    virtual mstate Model(real const &t, mstate const &X, int isJac) const;
    mstate ModelState(real const &t, mstate const &X) const;
    mstate ModelJacobian(real const &t, mstate const &X) const;
    virtual mcontrol Control(real const &t, mstate const &X) const;
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac) const;
};

#endif
