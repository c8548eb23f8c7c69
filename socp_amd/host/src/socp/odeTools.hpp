// odeTools.hpp -- host mirror of the reference's ODE toolbox interface (odeTools.hpp:18-192).
//
// Same class name, nested types, members and signatures, so user programs and models written
// against the reference compile unchanged.  What differs is where the work happens: integrate()
// hands the whole segment to the gfx950 kernels through the C-ABI (include/socp_hip.h) when the
// object is a model with device dynamics.  In-tree models always have them: for those there is no host loop and no GPU means
// an error.  A user class that overrides only the reference's host virtuals (DeviceModelId() == 0) is integrated by the
// reference's own loop on the host, with a one-line warning -- the plugin surface keeps working, at CPU speed.
#ifndef SOCP_AMD_ODETOOLS_HPP_
#define SOCP_AMD_ODETOOLS_HPP_

#include "commonType.hpp"

#include <sstream>
#include <vector>

class odeTools
{
public:
    typedef std::vector<real> odeVector;

    // odeTools.hpp:30-41: functor handed to integrate(); calls the virtual dynamics
    struct modelStruct {
        odeTools *m_ode;
        int m_isJac;
        modelStruct(odeTools *ode, int isJac) : m_ode(ode), m_isJac(isJac) {}
        virtual void operator()(odeVector const &X, odeVector &dXdt, real const &t) const { dXdt = m_ode->Model(t, X, m_isJac); }
    };

    // odeTools.hpp:46-57: functor called after every step when tracing
    struct observerStruct {
        odeTools *m_ode;
        std::stringstream &m_file;
        observerStruct(odeTools *ode, std::stringstream &file) : m_ode(ode), m_file(file) {}
        virtual void operator()(odeVector const &X, double const t) const { m_ode->Trace(t, X, m_file); }
    };

    odeTools() : odeIntTol(1e-8) {}
    virtual ~odeTools() {}

    real odeIntTol;   // tolerance of the adaptive integrator (odeTools.hpp:65)

    virtual void SetODEIntPrecision(real const &xtol) { odeIntTol = xtol; }

    // The reference picks its integrator at compile time: fixed-step RK4, or Boost.Odeint's adaptive
    // Dormand-Prince 5(4) with abs = rel = odeIntTol when built with -D_USE_BOOST (odeTools.cpp:11-15,129-134).
    // Here the same macro selects the default and the choice can also be made at run time (addition).
    static void UseAdaptiveIntegrator(bool on);
    static bool AdaptiveIntegrator();

    // dX/dt = Model(t, X); isJac = 1 asks for the variational system (odeTools.hpp:82)
    virtual odeVector Model(real const &t, odeVector const &X, int isJac = 0) const = 0;
    virtual void Trace(real const &t, odeVector const &X, std::stringstream &file) const = 0;

    static odeVector MultState(real a, odeVector const &X);
    static odeVector AddState(odeVector const &X, odeVector const &Y);

    // One-step helpers of the reference API (odeTools.hpp:107-165; operation order of odeTools.cpp:46-98).  They take
    // arbitrary host callbacks and run on the host (interceptor.cpp:117 uses the function-pointer form).
    static odeVector RK1(real const &t, odeVector const &X, real const &step, odeVector (*function)(real const &, odeVector const &, void *), void *context);
    static void RK1(real const &t, odeVector &X, real const &step, modelStruct const &ode);
    static odeVector RK2(real const &t, odeVector const &X, real const &step, odeVector (*function)(real const &, odeVector const &, void *), void *context);
    static void RK2(real const &t, odeVector &X, real const &step, modelStruct const &ode);
    static odeVector RK4(real const &t, odeVector const &X, real const &step, odeVector (*function)(real const &, odeVector const &, void *), void *context);
    static void RK4(real const &t, odeVector &X, real const &step, modelStruct const &ode);

    // odeTools.hpp:177,187: X(t0) -> X(tf) with step dt; the observer form also returns the
    // state after every step to the observer (trace).
    void integrate(modelStruct const &_model, odeVector &X, double const &t0, double const &tf, double const &dt, observerStruct const &_observer);
    void integrate(modelStruct const &_model, odeVector &X, double const &t0, double const &tf, double const &dt);
};

#endif
