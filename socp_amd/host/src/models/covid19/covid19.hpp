// covid19.hpp -- host mirror of the reference's SEIR epidemic model class (covid19.hpp:17-88);
// dynamics on the device (SOCP_MODEL_COVID19).
#ifndef SOCP_AMD_COVID19_HPP_
#define SOCP_AMD_COVID19_HPP_

#include "../../socp/model.hpp"
#include "../../socp/map.hpp"

#include <iostream>

class covid19 : public model
{
public:
    struct parameters_struct {
        real R0;     // secondary infections per infected individual
        real Tinf;   // infectious period
        real Tinc;   // incubation period
        real N;      // population size (1 when normalised)
        real Imax;   // tolerated infectious fraction
        real muI;    // penalty weight on I > Imax
        real umin;   // control bounds
        real umax;
    };

    covid19(std::string the_fileTrace = std::string(""));
    virtual ~covid19();
    parameters_struct &GetParameterData();

    virtual int DeviceModelId() const;
    virtual int DeviceParams(double *out, int cap) const;
    virtual int DeviceStepNumber() const;

private:
    struct data_struct;
    data_struct *data;

    virtual mstate Model(real const &t, mstate const &X, int isJac = 0) const;
    virtual mcontrol Control(real const &t, mstate const &X) const;
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac = 0) const;
    virtual mstate ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac = 0);
};

#endif
