// doubleIntegrator.hpp -- host mirror of the reference's 3-D double integrator model class
// (doubleIntegrator.hpp:15-81); dynamics on the device (SOCP_MODEL_DOUBLE_INTEGRATOR).
#ifndef SOCP_AMD_DOUBLEINTEGRATOR_HPP_
#define SOCP_AMD_DOUBLEINTEGRATOR_HPP_

#include <iostream>   // user programs written for the reference rely on these transitive includes

#include "../../socp/map.hpp"
#include "../../socp/model.hpp"

class doubleIntegrator : public model
{
    struct data_struct;
    data_struct *data;

    mstate Model(real const &t, mstate const &X, int isJac) const override;
    mcontrol Control(real const &t, mstate const &X) const override;
    mstate Hamiltonian(real const &t, mstate const &X, int isJac) const override;
    mstate ModelState(real const &t, mstate const &X) const;        // isJac = 0 / 1 halves of Model()
    mstate ModelJacobian(real const &t, mstate const &X) const;

public:
    struct parameters_struct {
        real u_max, a_max;   // control bound; acceleration scale
        real muT;            // weight of time in the cost
    };

    doubleIntegrator(int modelOrder, std::string the_fileTrace);
    ~doubleIntegrator() override;

    struct odeStruct;                // public in the reference too (doubleIntegrator.hpp:40-41)
    odeStruct *my_odeStruct;

    parameters_struct &GetParameterData();
    void SetStepNumber(int step);    // stored, never read -- as in the reference (doubleIntegrator.cpp:313-315)

    // device hook
    int DeviceModelId() const override;
    int DeviceParams(double *out, int cap) const override;
};

#endif
